// sgm_engine.hip -- host side of the C ABI declared in include/sgm_hip.h.
//
// Owns the device buffers and the stage schedule of one stereo matcher on one GPU / stream.
// Mirrors the reference's call shape (cv2.StereoSGBM_create -> .compute -> reprojectImageTo3D,
// /root/reference/main.ipynb:655-670, 697); see the header for the per-entry-point mapping.
#include "../../include/sgm_hip.h"
#include "sgm_debug.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kernels_cost.h"
#include "kernels_path.h"
#include "kernels_post.h"
#include "kernels_rectify.h"
#include "kernels_sweep.h"
#include "kernels_group.h"

using namespace sgm;

// ---- error plumbing ----------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return set_err(SGM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                           __FILE__, __LINE__);                                                    \
    } while (0)

// ---- engine --------------------------------------------------------------------------------
// Guarded allocation mode (environment SGM_DEBUG_ALLOC=1, read once per process; tests/test_gpu_guard.py runs parity
// cases under it in a child process).  Every device buffer of an engine is then placed through the HIP virtual-memory
// calls so that
//   * it ENDS exactly at the end of its mapping (sizes that are no multiple of 16 bytes: up to 15 bytes earlier) and the
//     pages behind -- and in front of -- the mapping are reserved but never mapped: a read or write past the buffer is a
//     GPU memory access fault at once instead of a silent touch of a neighbouring allocation.  Scalar loads (k_pix's
//     left-pixel records) and flat / global accesses carry no bounds check, unlike the buffer-descriptor accesses;
//   * the low half of every address inside the buffer has bit 31 SET: a 64-bit pointer put together from two 32-bit
//     halves with a signed low half (the int that __builtin_amdgcn_readfirstlane returns) turns into 0xffffffff'xxxxxxxx
//     there -- the cause of round 3's memory access fault, DESIGN.md 4.6 -- while hipMalloc hands out such addresses
//     only now and then.
// Buffers above 1 GiB keep plain hipMalloc (the tests that use the mode run small frames).
static int debug_alloc_mode()
{
    static const int m = [] {
        const char *s = getenv("SGM_DEBUG_ALLOC");
        return s ? atoi(s) : 0;
    }();
    return m;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // guarded mode only: the address range reserved, the part of it that is mapped, the physical allocation
    void *va = nullptr, *map = nullptr;
    size_t va_bytes = 0, map_bytes = 0;
    hipMemGenericAllocationHandle_t handle{};
    bool guarded = false;

    int ensure_guarded(size_t bytes)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return set_err(SGM_ERR_HIP, "hipGetDevice failed");
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0)
            return set_err(SGM_ERR_HIP, "SGM_DEBUG_ALLOC: hipMemGetAllocationGranularity failed");
        const size_t G4 = (size_t)1 << 32;
        const size_t mb = (bytes + gran - 1) / gran * gran;   // mapped bytes: whole granules
        // room for a 4 GiB boundary with the mapping right below it and one unmapped granule on either side
        const size_t vb = mb + G4 + 2 * gran;
        void *r = nullptr;
        if (hipMemAddressReserve(&r, vb, gran, nullptr, 0) != hipSuccess)
            return set_err(SGM_ERR_NOMEM, "SGM_DEBUG_ALLOC: hipMemAddressReserve(%zu) failed", vb);
        const uintptr_t r0 = (uintptr_t)r;
        const uintptr_t m1 = (r0 + gran + mb + G4 - 1) / G4 * G4;   // first 4 GiB boundary with room for guard + mapping below it
        const uintptr_t m0 = m1 - mb;                               // low halves of [m0, m1): [2^32 - mb, 2^32), bit 31 set (mb <= 2 GiB)
        hipMemGenericAllocationHandle_t h{};
        if (m0 < r0 + gran || m1 + gran > r0 + vb || hipMemCreate(&h, mb, &prop, 0) != hipSuccess) {
            (void)hipMemAddressFree(r, vb);
            return set_err(SGM_ERR_NOMEM, "SGM_DEBUG_ALLOC: hipMemCreate(%zu) failed", mb);
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemMap((void *)m0, mb, 0, h, 0) != hipSuccess || hipMemSetAccess((void *)m0, mb, &acc, 1) != hipSuccess) {
            (void)hipMemRelease(h);
            (void)hipMemAddressFree(r, vb);
            return set_err(SGM_ERR_HIP, "SGM_DEBUG_ALLOC: hipMemMap / hipMemSetAccess failed");
        }
        va = r;
        va_bytes = vb;
        map = (void *)m0;
        map_bytes = mb;
        handle = h;
        guarded = true;
        p = (void *)((m1 - bytes) & ~(uintptr_t)15);   // the buffer ends where the mapping ends (16-byte aligned start)
        cap = bytes;
        return SGM_OK;
    }
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SGM_OK;
        if (p) {
            if (release() != hipSuccess) return SGM_ERR_HIP;
        }
        if (debug_alloc_mode() && bytes <= ((size_t)1 << 30)) return ensure_guarded(bytes);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            set_err(SGM_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return SGM_ERR_NOMEM;
        }
        cap = bytes;
        return SGM_OK;
    }
    hipError_t release()
    {
        hipError_t e = hipSuccess;
        if (guarded) {
            e = hipMemUnmap(map, map_bytes);
            (void)hipMemRelease(handle);
            (void)hipMemAddressFree(va, va_bytes);
            guarded = false;
            va = map = nullptr;
        } else if (p) {
            e = hipFree(p);
        }
        p = nullptr;
        cap = 0;
        return e;
    }
};

// page-locked host memory (staging of sgm_compute_batch): copies to / from it are truly asynchronous
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SGM_OK;
        release();
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            set_err(SGM_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return SGM_ERR_NOMEM;
        }
        cap = bytes;
        return SGM_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct sgm_engine {
    sgm_params params;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t aux = nullptr;            // second stream: MODE_HH overlaps the upward pre-pass with the downward sweep
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t aux2 = nullptr;           // third stream (D <= 64, MODE_SGBM: the left-to-right in-row path beside everything else)
    hipEvent_t ev_join2 = nullptr;
    int keep_aggr = 0;
    int profile = 0;
    int schedule = 1;    // 0: one kernel per direction (v1); 1: fused 4-direction sweeps; 2: chained sweeps, no pre-pass (throughput mode)
    int chain_wgs = 0;   // schedule 2: workgroups (= bands in flight) per sweep launch; 0 = automatic
    bool plan_chain = false;  // what PH_PRE of the last compute decided: chained sweeps? band height, bands
    int plan_R = 0, plan_nbands = 0;
    int sweep_rows = 0;  // rows per band of the sweep (0 = automatic)
    int debug = 0;       // timing experiments (SweepArgs::dbg)
    int prepass_rows = 0;  // rows per chunk of the boundary pre-pass (0 = automatic, about 135, a multiple of 8)
    // sgm_compute_batch: up to three pairs in flight = this engine + two peers (own stream and device
    // buffers), each with page-locked staging buffers for the images and the disparity map
    sgm_engine *peer = nullptr, *peer2 = nullptr;
    std::vector<sgm_engine *> group;      // sgm_pipeline_batch_device: the other engines of a chained group (own streams and buffers)
    int last_group = 0;                   // engines of `group` the last batch call used (sgm_get_headroom looks at all of them)
    bool hr_accumulate = false;           // batch calls: the headroom record of this engine is NOT reset by the next compute (it then covers every pair the engine ran in the call)
    int group_max = 0;                    // SGM_OPT_GROUP_MAX: pairs per chained launch (0 = as many as fit in memory, up to CHAIN_MAX_FRAMES)
    // sgm_compute_batch, throughput mode: two groups in flight (the transfers of group g + 1 / g - 1 beside the kernels of
    // group g).  Each engine of a group keeps the device images of ITS pair twice -- slot g & 1: left, right, int16 map,
    // float map, XYZ -- with an event behind the upload and one behind the pair's last kernel; the engine the caller
    // holds owns the two copy streams.
    DevBuf io[2][5];
    HostBuf pin_io[2][3];                 // page-locked staging of the slot's images (left, right) and of its int16 map
    // events of a slot: images uploaded / images consumed by the cost stage / last kernel of the pair done / map downloaded
    hipEvent_t ev_io_in[2] = {nullptr, nullptr}, ev_io_used[2] = {nullptr, nullptr}, ev_io_out[2] = {nullptr, nullptr},
               ev_io_dl[2] = {nullptr, nullptr};
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    hipEvent_t ev_group = nullptr;
    HostBuf pin_left, pin_right, pin_disp;
    hipEvent_t ev_done = nullptr;

    // shape of the last compute
    int H = 0, W = 0;
    Geom g{};

    DevBuf in_left, in_right;           // staging for host-pointer calls
    DevBuf lrec, rplanes;               // features
    DevBuf hsum, cost, aggr;            // int16 [H][W1][D] volumes
    DevBuf aggr2;                       // MODE_SGBM, D <= 128: the fifth path's own volume (added to S by the winner-take-all)
    DevBuf aggr3;                       // MODE_SGBM, D <= 64: the other in-row path's own volume
    DevBuf aggr4, aggr5;                // MODE_SGBM, D <= 64: the volumes of the vertical and the second diagonal direction (k_paths5_g)
    DevBuf wta;                         // uint2 [H][W]
    DevBuf bndL, bndL2;                 // band-boundary state of the sweep pre-pass (down / up)
    DevBuf pstate, pstate2;             // line state between the row chunks of the pre-pass (ping-pong, down / up)
    DevBuf disp_raw, disp_med, disp_out;  // int16 [H][W]
    DevBuf label, csize, rlen;          // int32 [H][W] each
    DevBuf f32, xyz, mask, minkey;      // host-pointer post stages
    DevBuf rmap1, rmap2, rsrc, rdst;    // host-pointer rectification stages
    DevBuf ccount, cpts, crgb, crgb_in; // point compaction
    DevBuf headroom;                    // uint32[2]: max C_true (incl. upstream's running-sum intermediate), max min_d L_r
    DevBuf chain_ctl, chain_err;        // chained sweeps: ticket + progress words (zeroed before every launch); sticky give-up flag

    // profiling
    std::vector<hipEvent_t> events;
    std::vector<const char *> stage_names;
    std::vector<int> stage_launches;
    std::vector<uint64_t> stage_range;   // roctx range ids of the open stages
    int nstages = 0;
    int nevents = 0;                      // events of the pool used by this compute
    std::vector<int> stage_ev;            // [2 i], [2 i + 1]: begin / end event of stage i (indices into `events`)
    int last_end_ev = -1;                 // end event of the stage that was closed last, and its stream
    hipStream_t last_end_stream = nullptr;
};

static int normalise(const sgm_params *p, int H, int W, Geom *g)
{
    if (p->numDisparities <= 0) return set_err(SGM_ERR_INVALID_ARG, "numDisparities must be > 0");
    if (p->mode != 0 && p->mode != 1)
        return set_err(SGM_ERR_UNSUPPORTED, "mode %d: only MODE_SGBM (0) and MODE_HH (1) are built; the reference never selects 3WAY/HH4", p->mode);
    if (p->numDisparities % 16 != 0)
        return set_err(SGM_ERR_UNSUPPORTED, "numDisparities=%d must be divisible by 16 (OpenCV's documented contract)", p->numDisparities);
    if (p->numDisparities > 512) return set_err(SGM_ERR_UNSUPPORTED, "numDisparities=%d > 512", p->numDisparities);
    const int dim = p->blockSize > 0 ? p->blockSize : 5;
    if (dim > 31) return set_err(SGM_ERR_UNSUPPORTED, "blockSize=%d > 31", dim);
    g->H = H;
    g->W = W;
    g->minD = p->minDisparity;
    g->D = p->numDisparities;
    const int maxD = g->minD + g->D;
    g->minX1 = std::max(maxD, 0);
    const int maxX1 = W + std::min(g->minD, 0);
    g->W1 = maxX1 - g->minX1;
    g->SW2 = g->SH2 = dim / 2;
    g->P1 = p->P1 > 0 ? p->P1 : 2;
    g->P2 = std::max(p->P2 > 0 ? p->P2 : 5, g->P1 + 1);
    g->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    g->d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    g->ftzero = std::max(p->preFilterCap, 15) | 1;
    g->invalid_scaled = (g->minD - 1) * 16;
    g->mode = p->mode;
    g->NP = g->D <= 128 ? 1 : (g->D <= 256 ? 2 : 4);
    g->rowsz = (int64_t)std::max(g->W1, 0) * g->D;
    g->hr = nullptr;
    return SGM_OK;
}

// ---- stage bookkeeping -----------------------------------------------------------------------
// roctx ranges around every stage while SGM_OPT_PROFILE is on (named ranges in rocprofv3
// --marker-trace timelines; they bracket the host-side enqueue of the stage).  The tracing library
// is looked up at run time: no link-time dependency, silently absent when it is not installed.
struct Roctx {
    uint64_t (*start)(const char *) = nullptr;
    void (*stop)(uint64_t) = nullptr;
    Roctx()
    {
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_LOCAL)) {
                start = (uint64_t(*)(const char *))dlsym(h, "roctxRangeStartA");
                stop = (void (*)(uint64_t))dlsym(h, "roctxRangeStop");
                if (start && stop) return;
                start = nullptr;
                stop = nullptr;
            }
        }
    }
};
static Roctx &roctx()
{
    static Roctx r;
    return r;
}

// Stage brackets (SGM_OPT_PROFILE).  An event record is a marker packet that the next kernel of the stream
// waits for; two of them between every pair of stages cost a 720p frame 13 % (10-18 us per stage boundary
// on the rocprof timeline).  So a stage that begins right where the previous stage of the same stream
// ended -- nothing enqueued in between -- takes that stage's end event as its begin: one record per boundary.
static int new_stage_event(sgm_engine *e, hipStream_t on, int *idx)
{
    if ((size_t)e->nevents == e->events.size()) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->events.push_back(ev);
    }
    *idx = e->nevents++;
    HIP_TRY(hipEventRecord(e->events[*idx], on));
    return SGM_OK;
}
static int stage_begin(sgm_engine *e, const char *name, hipStream_t on = nullptr)
{
    if (!e->profile) return SGM_OK;
    hipStream_t st = on ? on : e->stream;
    const size_t i = (size_t)e->nstages;
    e->stage_range.resize(i + 1);
    if (roctx().start) e->stage_range[i] = roctx().start(name);
    e->stage_names.resize(i + 1);
    e->stage_launches.resize(i + 1);
    e->stage_ev.resize(2 * (i + 1));
    e->stage_names[i] = name;
    e->stage_launches[i] = 0;
    if (e->last_end_ev >= 0 && e->last_end_stream == st) {
        e->stage_ev[2 * i] = e->last_end_ev;
    } else {
        int rc = new_stage_event(e, st, &e->stage_ev[2 * i]);
        if (rc) return rc;
    }
    return SGM_OK;
}
static int stage_end(sgm_engine *e, int launches, hipStream_t on = nullptr)
{
    if (!e->profile) return SGM_OK;
    hipStream_t st = on ? on : e->stream;
    const size_t i = (size_t)e->nstages;
    e->stage_launches[i] = launches;
    int rc = new_stage_event(e, st, &e->stage_ev[2 * i + 1]);
    if (rc) return rc;
    e->last_end_ev = e->stage_ev[2 * i + 1];
    e->last_end_stream = st;
    if (roctx().stop) roctx().stop(e->stage_range[i]);
    e->nstages++;
    return SGM_OK;
}
// something other than a stage was enqueued (a wait for another stream, a memset ...): the next stage needs a begin event of its own
static void stage_break(sgm_engine *e) { e->last_end_ev = -1; }

#define KCHECK() HIP_TRY(hipGetLastError())

// ---- path launch dispatch ---------------------------------------------------------------------
template <int NP, bool PARTIAL>
static void launch_path_np(const Geom &g, int rx, int ry, int mode, const int16_t *C, int16_t *S, int keepS,
                           uint2 *wta, Boundary bd, hipStream_t st)
{
    const int nlines = ry == 0 ? g.H : g.W1;
    dim3 grid(nlines, mode == PATH_BOUNDARY ? 3 : 1), block(64);
    const bool posw = g.uniq < 100;  // positive uniqueness weight: the WTA variant without products
    if (mode == PATH_FIRST)
        hipLaunchKernelGGL((k_path<NP, PARTIAL, PATH_FIRST, true>), grid, block, 0, st, g, rx, ry, C, S, keepS, wta, bd);
    else if (mode == PATH_ACCUM)
        hipLaunchKernelGGL((k_path<NP, PARTIAL, PATH_ACCUM, true>), grid, block, 0, st, g, rx, ry, C, S, keepS, wta, bd);
    else if (mode == PATH_LAST && posw)
        hipLaunchKernelGGL((k_path<NP, PARTIAL, PATH_LAST, true>), grid, block, 0, st, g, rx, ry, C, S, keepS, wta, bd);
    else if (mode == PATH_LAST)
        hipLaunchKernelGGL((k_path<NP, PARTIAL, PATH_LAST, false>), grid, block, 0, st, g, rx, ry, C, S, keepS, wta, bd);
    else
        hipLaunchKernelGGL((k_path<NP, PARTIAL, PATH_BOUNDARY, true>), grid, block, 0, st, g, rx, ry, C, S, keepS, wta, bd);
}

static void launch_path(const Geom &g, int rx, int ry, int mode, const int16_t *C, int16_t *S, int keepS,
                        uint2 *wta, hipStream_t st, Boundary bd = Boundary{nullptr, 1, 0})
{
    const bool partial = g.D != 128 * g.NP;
    if (g.NP == 1) {
        if (partial) launch_path_np<1, true>(g, rx, ry, mode, C, S, keepS, wta, bd, st);
        else launch_path_np<1, false>(g, rx, ry, mode, C, S, keepS, wta, bd, st);
    } else if (g.NP == 2) {
        if (partial) launch_path_np<2, true>(g, rx, ry, mode, C, S, keepS, wta, bd, st);
        else launch_path_np<2, false>(g, rx, ry, mode, C, S, keepS, wta, bd, st);
    } else {
        if (partial) launch_path_np<4, true>(g, rx, ry, mode, C, S, keepS, wta, bd, st);
        else launch_path_np<4, false>(g, rx, ry, mode, C, S, keepS, wta, bd, st);
    }
}

// ---- sweep launch dispatch --------------------------------------------------------------------
// ---- small D: lane-grouped kernels (kernels_group.h) ----
static int group_width(const Geom &g, int H)
{
    if (g.D > 64 || (int64_t)H * g.rowsz * 2 >= (int64_t)0x7fff0000) return 64;
    return g.D <= 16 ? 8 : (g.D <= 32 ? 16 : 32);
}
template <int GW, int NP, bool PARTIAL>
static void launch_rows_g(const Geom &g, int H, int rx, int mode, const int16_t *C, int16_t *S, int keepS, uint2 *wta, hipStream_t st)
{
    constexpr int G = 64 / GW;
    dim3 grid((H + G - 1) / G), block(64);
    if (mode == PATH_FIRST)
        hipLaunchKernelGGL((k_rows_g<GW, NP, PARTIAL, PATH_FIRST, true>), grid, block, 0, st, g, rx, C, S, keepS, wta);
    else if (mode == PATH_ACCUM)
        hipLaunchKernelGGL((k_rows_g<GW, NP, PARTIAL, PATH_ACCUM, true>), grid, block, 0, st, g, rx, C, S, keepS, wta);
    else if (g.uniq < 100)
        hipLaunchKernelGGL((k_rows_g<GW, NP, PARTIAL, PATH_LAST, true>), grid, block, 0, st, g, rx, C, S, keepS, wta);
    else
        hipLaunchKernelGGL((k_rows_g<GW, NP, PARTIAL, PATH_LAST, false>), grid, block, 0, st, g, rx, C, S, keepS, wta);
}
// the in-row path of a pass (rows are independent): GW < 64 packs 64/GW rows into a wave
static void launch_rows_grouped(const Geom &g, int H, int GW, int rx, int mode, const int16_t *C, int16_t *S, int keepS, uint2 *wta, hipStream_t st)
{
    const bool partial = g.D != 128 * g.NP;
    // lane groups: PARTIAL = the group is not full (D = 48 in groups of 32; D = 16, 32, 64 fill theirs)
    if (GW == 8) launch_rows_g<8, 1, false>(g, H, rx, mode, C, S, keepS, wta, st);
    else if (GW == 16) launch_rows_g<16, 1, false>(g, H, rx, mode, C, S, keepS, wta, st);
    else if (GW == 32) { if (g.D < 64) launch_rows_g<32, 1, true>(g, H, rx, mode, C, S, keepS, wta, st); else launch_rows_g<32, 1, false>(g, H, rx, mode, C, S, keepS, wta, st); }
    else if (g.NP == 1) { if (partial) launch_rows_g<64, 1, true>(g, H, rx, mode, C, S, keepS, wta, st); else launch_rows_g<64, 1, false>(g, H, rx, mode, C, S, keepS, wta, st); }
    else if (g.NP == 2) { if (partial) launch_rows_g<64, 2, true>(g, H, rx, mode, C, S, keepS, wta, st); else launch_rows_g<64, 2, false>(g, H, rx, mode, C, S, keepS, wta, st); }
    else { if (partial) launch_rows_g<64, 4, true>(g, H, rx, mode, C, S, keepS, wta, st); else launch_rows_g<64, 4, false>(g, H, rx, mode, C, S, keepS, wta, st); }
}

template <int NP, bool PARTIAL, int MODE, bool POSW>
static int launch_sweep_one(const Geom &g, const SweepArgs &a, int nbands, hipStream_t st)
{
    const size_t lds = sweep_lds_bytes(NP, a.R);
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweep<NP, PARTIAL, MODE, POSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_sweep<NP, PARTIAL, MODE, POSW>), dim3(nbands), dim3((a.R + 1) * 64), lds, st, g, a);
    return SGM_OK;
}
template <int NP, bool PARTIAL>
static int launch_sweep_np(const Geom &g, const SweepArgs &a, int mode, int nbands, hipStream_t st)
{
    if (mode == SWEEP_FIRST) return launch_sweep_one<NP, PARTIAL, SWEEP_FIRST, true>(g, a, nbands, st);
    if (mode == SWEEP_ACCUM) return launch_sweep_one<NP, PARTIAL, SWEEP_ACCUM, true>(g, a, nbands, st);
    if (g.uniq < 100) return launch_sweep_one<NP, PARTIAL, SWEEP_LAST, true>(g, a, nbands, st);
    return launch_sweep_one<NP, PARTIAL, SWEEP_LAST, false>(g, a, nbands, st);
}
// the sweep kernels take the direction of the walk from the pass (kernels_sweep.h: XD): the first pass runs top-down and left
// to right, every other pass the other way round
static int check_sweep_direction(const SweepArgs &a, int mode)
{
    if ((mode == SWEEP_FIRST) != (a.xdir > 0) || a.xdir != a.ydir)
        return set_err(SGM_ERR_INVALID_ARG, "internal: sweep mode %d with direction (%d, %d)", mode, a.xdir, a.ydir);
    return SGM_OK;
}
static int launch_sweep(const Geom &g, const SweepArgs &a, int mode, int nbands, hipStream_t st)
{
    if (int rc = check_sweep_direction(a, mode)) return rc;
    const bool partial = g.D != 128 * g.NP;
    if (g.NP == 1) return partial ? launch_sweep_np<1, true>(g, a, mode, nbands, st) : launch_sweep_np<1, false>(g, a, mode, nbands, st);
    if (g.NP == 2) return partial ? launch_sweep_np<2, true>(g, a, mode, nbands, st) : launch_sweep_np<2, false>(g, a, mode, nbands, st);
    return partial ? launch_sweep_np<4, true>(g, a, mode, nbands, st) : launch_sweep_np<4, false>(g, a, mode, nbands, st);
}

// chained sweep (kernels_sweep.h: k_sweep_chain): `wgs` persistent workgroups of R compute waves + loader + publisher
template <int NP, bool PARTIAL, int MODE>
static int launch_chain_one(const Geom &g, const SweepArgs &a, const ChainFrames &fr, int wgs, hipStream_t st)
{
    const size_t lds = sweep_lds_bytes(NP, a.R) + 16;  // + the ticket word
    HIP_TRY(hipFuncSetAttribute((const void *)k_sweep_chain<NP, PARTIAL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_sweep_chain<NP, PARTIAL, MODE>), dim3(wgs), dim3((a.R + 2) * 64), lds, st, g, a, fr);
    return SGM_OK;
}
template <int NP, bool PARTIAL>
static int launch_chain_np(const Geom &g, const SweepArgs &a, const ChainFrames &fr, int mode, int wgs, hipStream_t st)
{
    if (mode == SWEEP_FIRST) return launch_chain_one<NP, PARTIAL, SWEEP_FIRST>(g, a, fr, wgs, st);
    return launch_chain_one<NP, PARTIAL, SWEEP_ACCUM>(g, a, fr, wgs, st);
}
static int launch_chain(const Geom &g, const SweepArgs &a, const ChainFrames &fr, int mode, int wgs, hipStream_t st)
{
    if (int rc = check_sweep_direction(a, mode)) return rc;
    const bool partial = g.D != 128 * g.NP;
    if (g.NP == 1) return partial ? launch_chain_np<1, true>(g, a, fr, mode, wgs, st) : launch_chain_np<1, false>(g, a, fr, mode, wgs, st);
    if (g.NP == 2) return partial ? launch_chain_np<2, true>(g, a, fr, mode, wgs, st) : launch_chain_np<2, false>(g, a, fr, mode, wgs, st);
    return partial ? launch_chain_np<4, true>(g, a, fr, mode, wgs, st) : launch_chain_np<4, false>(g, a, fr, mode, wgs, st);
}
// workgroups of a chained launch over nf frames: a band trails the band above by about 2 (R - 1) + 17 lockstep steps and
// lasts T steps, so a frame keeps about T / lag workgroups busy; more would only wait (and hold CUs)
static int chain_window(const Geom &g, int R, int nbands, int nf, int override_wgs)
{
    const int pps = sweep_pps(g.NP);
    const int T = (g.W1 + pps - 1) / pps + 2 * (R - 1), lag = 2 * (R - 1) + 17;
    const int per_frame = override_wgs > 0 ? override_wgs : std::max(4, (T + lag - 1) / lag);
    return (int)std::min<int64_t>({(int64_t)per_frame * nf, (int64_t)nbands * nf, 256});
}

// rows per band: about 240 bands (one workgroup per CU, most of the 256 CUs busy), bounded by
// SWEEP_MAX_ROWS (register budget of the workgroup) and the 160 KiB of LDS.  MODE_HH: about 200 bands --
// a band's step time is set by its busiest SIMD (3 waves with 9 rows + loader as with 11 + loader), so
// taller bands cost the sweep nothing, write and read 18 % less boundary state and leave more CUs to the
// upward pre-pass that runs beside the downward sweep (4K: 11.05 -> 10.9 ms; MODE_SGBM: no gain).
// Chained schedule: bands are not tied to the number of CUs (a window of workgroups slides over them), so the tallest
// band the workgroup can hold: 12 rows = three compute waves on every SIMD, and the least hand-off traffic (3 / R volumes).
static int sweep_rows_for(const Geom &g, int override_rows, int npass, bool chained = false)
{
    int maxR = chained ? CHAIN_MAX_ROWS : SWEEP_MAX_ROWS;
    while (maxR > 1 && sweep_lds_bytes(g.NP, maxR) + 16 > 160 * 1024) maxR--;
    if (chained && override_rows <= 0) return std::max(1, std::min(maxR, g.H));
    const int bands = npass == 2 ? 200 : 240;
    int R = override_rows > 0 ? override_rows : (g.H + bands - 1) / bands;
    if (override_rows <= 0) R = std::max(R, 4);
    return std::max(1, std::min(R, maxR));
}

// ---- the schedule of one compute, decided from the geometry and the engine's options alone ----------------
// (run_compute follows it; the batch entries read it BEFORE anything is allocated or enqueued: whether a configuration
// runs chained, and what a pair costs in device memory)
struct Plan {
    bool byte_cost;   // per-pixel cost as bytes + k_box_u8 (else the int16 pipeline k_hsum + k_vsum*)
    int GWs;          // lane-group width of the small-D kernels (64: none)
    bool rows4;       // small-D schedule (D <= 64 outside throughput mode, D <= 32 always)
    bool chain;       // chained sweeps (schedule 2)
    int npass, R, nbands;
};
static Plan make_plan(const sgm_engine *e, const Geom &g, int H)
{
    Plan p;
    // Byte pipeline (default): k_pix writes the per-pixel cost as uint8, k_box_u8 does the whole box
    // filter from it.  Needs a window radius 1..5 (instantiations), a cost that fits a byte, and a
    // pix volume below the 2 GiB a buffer descriptor spans here.  D <= 64: k_box_u8 in lane groups (several columns'
    // disparities side by side in a wave); the per-pixel cost comes from k_pix_px (D <= 32, one thread per pixel) or
    // k_pix (D = 48, 64: half its lanes idle, still less than the int16 pipeline's 3 V more traffic).
    // debug 256: the int16 pipeline always; debug 4 (no lane groups): the int16 pipeline for D <= 64.
    p.byte_cost = !(e->debug & 256) && (g.D > 64 || !(e->debug & 4)) && g.SW2 >= 1 && g.SW2 <= 5 && g.SH2 == g.SW2 &&
                  2 * g.ftzero + 63 <= 255 && (int64_t)H * g.rowsz < (int64_t)0x7ff00000;
    // D <= 32: rows without hand-off: band height 1, the pre-pass stores every row's
    // state; needs the 3-volume state buffer below the 4 GiB a 32-bit buffer offset reaches.
    // (At D = 64 the fused sweep is still ahead: 720p 0.35 against 0.39 ms, and 3 V of state.)
    p.GWs = (e->debug & 4) ? 64 : group_width(g, H);
    // Throughput mode (schedule 2) takes D = 48 .. 64 through the chained sweeps all the same (half the lanes idle, but
    // 7 V of traffic per pair instead of the 22 V of the per-row state: batches of small frames are bound by HBM --
    // 64 pairs 720p D=64: 0.38 against 0.55 ms per pair); D <= 32 keeps the small-D kernels in every mode.
    p.rows4 = p.GWs <= 32 && e->sweep_rows <= 0 && !(e->schedule == 2 && p.GWs == 32) &&
              (int64_t)H * g.rowsz * 2 * 3 < (int64_t)0xfff00000;
    p.npass = g.mode == 1 ? 2 : 1;
    // Chained schedule (SGM_OPT_SCHEDULE 2, kernels_sweep.h: k_sweep_chain): no pre-pass; the bands of a sweep hand the
    // state of their last row to each other.  Only where the fused sweep runs (the small-D schedule keeps its own
    // kernels), where there is more than one band, and not with debug 2 (winner-take-all inside the second sweep).
    p.chain = e->schedule == 2 && !p.rows4 && !((e->debug & 2) && g.mode == 1) && g.W1 > 0;
    p.R = p.rows4 ? 1 : sweep_rows_for(g, e->sweep_rows, p.npass, p.chain);
    if (p.chain && (H + p.R - 1) / p.R <= 1) {
        p.chain = false;
        p.R = sweep_rows_for(g, e->sweep_rows, p.npass, false);
    }
    p.nbands = (H + p.R - 1) / p.R;
    return p;
}

// ---- the matcher on device buffers ---------------------------------------------------------
static int ensure_buffers(sgm_engine *e, int H, int W)
{
    Geom g;
    int rc = normalise(&e->params, H, W, &g);
    if (rc) return rc;
    e->g = g;
    e->H = H;
    e->W = W;
    const size_t npx = (size_t)H * W;
    const size_t vol = (size_t)std::max<int64_t>(g.rowsz, 0) * H * sizeof(int16_t);
    // 64 bytes of slack behind the last record: k_pix reads the records with scalar loads -- no bounds check, and the
    // compiler may merge or widen them (the widest scalar load is 64 bytes).  Every load STARTS at a record of the
    // frame, so none can leave the allocation.  (Not in the guarded mode: there the buffer ends where its mapping
    // ends, and the parity cases of tests/test_gpu_guard.py show that no load goes past the last record at all.)
    if ((rc = e->lrec.ensure(npx * 8 + (debug_alloc_mode() ? 0 : 64)))) return rc;
    if ((rc = e->rplanes.ensure(npx * 6))) return rc;
    if (vol) {
        // The byte pipeline keeps its per-pixel costs (V / 2) in the S buffer: they are dead when the block cost C is
        // complete, and no kernel writes S before that (the sweeps, the in-row paths and k_paths5_g all read C; in a batch
        // every pair's cost stage is complete before the joint sweep launch starts).  Only the int16 pipeline needs a
        // volume of its own for the horizontal sums.  4K D=256: 13 -> 9 GB per engine (round 3 allocated V for them always).
        if (!make_plan(e, g, H).byte_cost && (rc = e->hsum.ensure(vol))) return rc;
        if ((rc = e->cost.ensure(vol))) return rc;
        if ((rc = e->aggr.ensure(vol))) return rc;
    }
    if ((rc = e->wta.ensure(npx * 8))) return rc;
    if ((rc = e->disp_raw.ensure(npx * 2))) return rc;
    if ((rc = e->disp_med.ensure(npx * 2))) return rc;
    if ((rc = e->headroom.ensure(8))) return rc;
    e->g.hr = (uint32_t *)e->headroom.p;
    return SGM_OK;
}

static int run_speckles(sgm_engine *e, int16_t *d_img, int H, int W, int newVal, int maxSpeckleSize, int maxDiff)
{
    const size_t npx = (size_t)H * W;
    int rc;
    if ((rc = e->label.ensure(npx * 4)) || (rc = e->csize.ensure(npx * 4)) || (rc = e->rlen.ensure(npx * 4))) return rc;
    int *label = (int *)e->label.p, *csz = (int *)e->csize.p, *rlen = (int *)e->rlen.p;
    hipStream_t st = e->stream;
    dim3 g2((W + 255) / 256, H);
    hipLaunchKernelGGL(k_ccl_rows, dim3(H), dim3(64), 0, st, (const int16_t *)d_img, label, rlen, csz, W, newVal, maxDiff);
    hipLaunchKernelGGL(k_ccl_merge, g2, dim3(256), 0, st, (const int16_t *)d_img, label, H, W, newVal, maxDiff);
    hipLaunchKernelGGL(k_ccl_count, g2, dim3(256), 0, st, (const int16_t *)d_img, label, (const int *)rlen, csz, H, W, newVal, maxDiff);
    hipLaunchKernelGGL(k_ccl_apply, g2, dim3(256), 0, st, d_img, label, (const int *)csz, H, W, newVal, maxDiff, maxSpeckleSize);
    KCHECK();
    return SGM_OK;
}

__global__ void k_fill_i16(int16_t *p, int64_t n, int16_t v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Phases of one compute.  A single pair runs them all; the batch entry with chained sweeps
// (sgm_pipeline_batch_device) runs PH_PRE of every pair, then ONE sweep launch per pass for all pairs, then PH_POST
// of every pair.
enum { PH_PRE = 1 /* features, block cost, MODE_SGBM's fifth path beside the sweep */, PH_MID = 2 /* pre-pass + sweeps */,
       PH_POST = 4 /* the rest */, PH_ALL = 7 };

static int run_compute(sgm_engine *e, const uint8_t *d_left, const uint8_t *d_right, int H, int W,
                       int64_t stride, int16_t *d_disp, int phases = PH_ALL)
{
    if (!e || !d_left || !d_right || !d_disp) return set_err(SGM_ERR_INVALID_ARG, "null pointer");
    if (H <= 0 || W < 2 || stride < W) return set_err(SGM_ERR_INVALID_ARG, "bad shape H=%d W=%d stride=%lld", H, W, (long long)stride);
    if (W > 32767 || H > 32767) return set_err(SGM_ERR_UNSUPPORTED, "image larger than 32767 in a dimension");
    HIP_TRY(hipSetDevice(e->device));
    int rc = ensure_buffers(e, H, W);
    if (rc) return rc;
    const Geom &g = e->g;
    hipStream_t st = e->stream;
    const bool do_pre = (phases & PH_PRE) != 0, do_mid = (phases & PH_MID) != 0, do_post = (phases & PH_POST) != 0;
    if (do_pre) {
        e->nstages = 0;
        e->nevents = 0;
        e->last_end_ev = -1;
        e->plan_chain = false;
    } else {
        stage_break(e);
    }
    const int64_t npx = (int64_t)H * W;
    const unsigned nb_px = (unsigned)((npx + 255) / 256);

    int16_t *raw = (int16_t *)e->disp_raw.p, *med = (int16_t *)e->disp_med.p;
    if (do_pre && !e->hr_accumulate) HIP_TRY(hipMemsetAsync(e->headroom.p, 0, 8, st));  // headroom record of this compute (sgm_get_headroom)

    if (g.W1 <= 0) {
        // no column can be matched: the whole map is invalid (upstream early-out), then median
        // and speckle act on a constant image
        if (phases != PH_ALL) return set_err(SGM_ERR_INVALID_ARG, "phased compute on a frame without matchable columns");
        if ((rc = stage_begin(e, "fill_invalid"))) return rc;
        hipLaunchKernelGGL(k_fill_i16, dim3(nb_px), dim3(256), 0, st, raw, npx, (int16_t)g.invalid_scaled);
        KCHECK();
        if ((rc = stage_end(e, 1))) return rc;
    } else {
        const int16_t *C = (const int16_t *)e->cost.p;
        int16_t *S = (int16_t *)e->aggr.p;
        int16_t *HS = (int16_t *)e->hsum.p;
        uint2 *wta = (uint2 *)e->wta.p;

        // -- features
        if (do_pre) {
            if ((rc = stage_begin(e, "features"))) return rc;
            dim3 grid((W + 255) / 256, H, 2), block(256);
            hipLaunchKernelGGL(k_features, grid, block, 0, st, d_left, d_right, stride, H, W, g.ftzero, (uint2 *)e->lrec.p,
                               (uint8_t *)e->rplanes.p);
            KCHECK();
            if ((rc = stage_end(e, 1))) return rc;
        }

        // -- horizontal box sum of the pixel cost
        // -- horizontal box sum of the pixel cost (rows y0 .. y1-1), vertical box sum -> block cost
        const int XL = 128;
        const int nchunks = (g.W1 + XL - 1) / XL;
        int RS = 1;  // ring of the last blockSize+1 cost vectors, rounded to a power of two
        while (RS < 2 * g.SW2 + 2) RS <<= 1;
        const HsumLds l = hsum_lds_layout(g.NP, RS, XL, g.SW2);
        const uint2 *lrec = (const uint2 *)e->lrec.p;
        const uint8_t *rpl = (const uint8_t *)e->rplanes.p;
        auto launch_hsum = [&](int hs_y0, int hs_y1) -> int {
            dim3 grid((unsigned)((int64_t)(hs_y1 - hs_y0) * nchunks)), block(64);
            // RS_T = RS instantiations carry the unrolled interior fast path (block sizes up to 15)
#define SGM_HSUM(NP_, RS_)                                                                                          \
    do {                                                                                                            \
        if (l.total_bytes > 48 * 1024)                                                                              \
            HIP_TRY(hipFuncSetAttribute((const void *)k_hsum<NP_, RS_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        l.total_bytes));                                                            \
        hipLaunchKernelGGL((k_hsum<NP_, RS_>), grid, block, l.total_bytes, st, g, lrec, rpl, HS, XL, nchunks, RS,    \
                           l.ring_bytes, l.lrec_bytes, l.seg_len, hs_y0);                                           \
    } while (0)
#define SGM_HSUM_NP(NP_)                    \
    do {                                    \
        if (RS == 4) SGM_HSUM(NP_, 4);      \
        else if (RS == 8) SGM_HSUM(NP_, 8); \
        else if (RS == 16) SGM_HSUM(NP_, 16); \
        else SGM_HSUM(NP_, 0);              \
    } while (0)
            if (g.NP == 1) SGM_HSUM_NP(1);
            else if (g.NP == 2) SGM_HSUM_NP(2);
            else SGM_HSUM_NP(4);
#undef SGM_HSUM_NP
#undef SGM_HSUM
            return SGM_OK;
        };
        const int RB = 96;  // rows per band of the vertical sum; multiple of every ring size used below
        const int nvb = (H + RB - 1) / RB;
        const int16_t *hsp = (const int16_t *)HS;
        int16_t *cp = (int16_t *)e->cost.p;
        auto launch_vsum = [&](int nb) {
            const bool wide = !(e->debug & 8);  // 8 int16 per thread (debug 8: 4, for A/B timing)
            const int per_thread = wide ? 8 : 4;
            dim3 block(256), gridr((unsigned)((g.rowsz / per_thread + 255) / 256), nb);
#define SGM_VSUM(SH2_)                                                                                             \
    case SH2_:                                                                                                     \
        if (wide) hipLaunchKernelGGL((k_vsum_ring<SH2_, 4>), gridr, block, 0, st, hsp, cp, H, g.rowsz, RB, g.hr); \
        else hipLaunchKernelGGL((k_vsum_ring<SH2_, 2>), gridr, block, 0, st, hsp, cp, H, g.rowsz, RB, g.hr); \
        break;
            switch (g.SH2) {  // ring variant: each hsum row is read once
                SGM_VSUM(1)
                SGM_VSUM(2)
                SGM_VSUM(3)
                SGM_VSUM(4)
                SGM_VSUM(5)
#undef SGM_VSUM
            default: break;
            }
        };
        const Plan plan = make_plan(e, g, H);   // (the same decisions the batch entries read before they enqueue anything)
        const bool byte_cost = plan.byte_cost;
        const int GWc = g.D > 64 ? 64 : (g.D <= 16 ? 8 : (g.D <= 32 ? 16 : 32));  // lane-group width of k_box_u8
        const int cpw = 16 * (64 / GWc);  // columns per workgroup of k_box_u8: 4 waves x (64 / GW groups) x 4 columns
        // rows per band of k_box_u8 (a band re-reads 2 * SH2 rows above it; multiples of 16: the register rings): 96, less
        // on frames too small to fill the chip with bands that tall
        int RBb = 96;
        for (int cand : {48, 32, 16})
            if ((int64_t)((g.W1 + cpw - 1) / cpw) * ((H + RBb - 1) / RBb) * 4 < 1024) RBb = cand;
        const int nvbb = (H + RBb - 1) / RBb;
        // per-pixel cost of rows [y_lo, y_hi) / block cost of the nb bands of RB rows, on stream `on`
        auto launch_pix = [&](int y_lo, int y_hi, hipStream_t on) {
            const int nj = XL + 2;
            const int lrec_b = ((nj * 8) + 15) & ~15;
            const int seg_l = (nj + 128 * g.NP + 15) & ~15;
            const size_t lds = (size_t)lrec_b + 6 * (size_t)seg_l;
            dim3 grid((unsigned)((int64_t)(y_hi - y_lo) * nchunks)), block(64);
            uint8_t *px = (uint8_t *)e->aggr.p;   // (the byte costs live in the S buffer until C is complete: ensure_buffers)
            if (g.D <= 32) {  // one thread per pixel (whole frame; callers pass [0, H))
                dim3 gridp((g.W1 + 255) / 256, H), blockp(256);
                hipLaunchKernelGGL(k_pix_px, gridp, blockp, (size_t)6 * (256 + g.D), on, g, lrec, rpl, px);
            } else
            if (g.NP == 1) hipLaunchKernelGGL(k_pix<1>, grid, block, lds, on, g, lrec, rpl, px, XL, nchunks, lrec_b, seg_l, y_lo);
            else if (g.NP == 2) hipLaunchKernelGGL(k_pix<2>, grid, block, lds, on, g, lrec, rpl, px, XL, nchunks, lrec_b, seg_l, y_lo);
            else hipLaunchKernelGGL(k_pix<4>, grid, block, lds, on, g, lrec, rpl, px, XL, nchunks, lrec_b, seg_l, y_lo);
        };
        auto launch_box = [&](int nb, hipStream_t on) {
            dim3 grid((g.W1 + cpw - 1) / cpw, nb), block(256);
            const uint8_t *px = (const uint8_t *)e->aggr.p;
            int16_t *cp2 = (int16_t *)e->cost.p;
#define SGM_BOX(R_)                                                                                        \
    case R_:                                                                                               \
        if (GWc == 8) hipLaunchKernelGGL((k_box_u8<R_, 1, 8>), grid, block, 0, on, g, px, cp2, RBb);        \
        else if (GWc == 16) hipLaunchKernelGGL((k_box_u8<R_, 1, 16>), grid, block, 0, on, g, px, cp2, RBb); \
        else if (GWc == 32) hipLaunchKernelGGL((k_box_u8<R_, 1, 32>), grid, block, 0, on, g, px, cp2, RBb); \
        else if (g.NP == 1) hipLaunchKernelGGL((k_box_u8<R_, 1>), grid, block, 0, on, g, px, cp2, RBb);  \
        else if (g.NP == 2) hipLaunchKernelGGL((k_box_u8<R_, 2>), grid, block, 0, on, g, px, cp2, RBb); \
        else hipLaunchKernelGGL((k_box_u8<R_, 4>), grid, block, 0, on, g, px, cp2, RBb);            \
        break;
            switch (g.SW2) {
                SGM_BOX(1)
                SGM_BOX(2)
                SGM_BOX(3)
                SGM_BOX(4)
                SGM_BOX(5)
            default: break;
            }
#undef SGM_BOX
        };
        // ---- schedule parameters (make_plan)
        const int GWs = plan.GWs;
        const bool rows4 = plan.rows4;
        const int npass = plan.npass;
        const bool chain = plan.chain;
        const int R = plan.R;
        const int nbands = plan.nbands;
        // Narrow frames (fewer than ~1.5 lines per SIMD) are bound by the latency of one wave's
        // instruction stream: there the single-direction kernel with one wave per (line, role) -- three
        // times the waves, a third of the work each -- is faster (720p D=64: 0.26 against 0.34 ms); its
        // three readers of C are served by L2 / the Infinity Cache at these sizes.
        const bool narrow = g.W1 <= 1536 && e->prepass_rows == 0;  // (an explicit chunk height selects k_prepass3: tests)
        const bool fused_prepass = !rows4 && !(e->debug & 16) && !narrow && (int64_t)g.rowsz * H < (1ll << 31);
        // (Running the cost stage and the downward pre-pass as a pipeline over row chunks on separate
        // streams was built and measured in round 2: the overlapped kernels only slow each other down --
        // cost_box 1.17 -> 2.75 ms, prepass_dn 2.05 -> 3.19 ms, frame 11.97 against 11.90 ms -- this phase
        // of the frame is bound by HBM bandwidth, not by the order of its launches.  DESIGN.md 4.4.)
        if (!do_pre) {
            // (the cost stage belongs to PH_PRE)
        } else if (byte_cost) {
            if ((rc = stage_begin(e, "cost_pix"))) return rc;
            launch_pix(0, H, st);
            KCHECK();
            if ((rc = stage_end(e, 1))) return rc;
            if ((rc = stage_begin(e, "cost_box"))) return rc;
            launch_box(nvbb, st);
            KCHECK();
            if ((rc = stage_end(e, 1))) return rc;
        } else {
        if ((rc = stage_begin(e, "cost_hsum"))) return rc;
        if (g.D <= 32 && !(e->debug & 4) && 2 * g.ftzero + 63 <= 255) {
            // D <= 32: one thread per pixel (lanes spanning D would mostly idle; at D = 64 the wave-per-
            // chunk kernel is still ahead); the byte volume borrows the S buffer, unused before the paths
            uint8_t *px = (uint8_t *)e->aggr.p;
            dim3 grid((g.W1 + 255) / 256, H), block(256);
            hipLaunchKernelGGL(k_pix_px, grid, block, (size_t)6 * (256 + g.D), st, g, lrec, rpl, px);
            hipLaunchKernelGGL(k_hsum_px, grid, block, 0, st, g, (const uint8_t *)px, (int16_t *)HS);
        } else if ((rc = launch_hsum(0, H))) {
            return rc;
        }
        KCHECK();
        if ((rc = stage_end(e, 1))) return rc;
        if ((rc = stage_begin(e, "cost_vsum"))) return rc;
        if (g.SH2 >= 1 && g.SH2 <= 5) {
            launch_vsum(nvb);
        } else {
            dim3 block(256), grid((unsigned)((g.rowsz / 8 + 255) / 256), (H + 63) / 64);
            hipLaunchKernelGGL(k_vsum, grid, block, 0, st, hsp, cp, H, g.rowsz, g.SH2, 64, g.hr);
        }
        KCHECK();
        if ((rc = stage_end(e, 1))) return rc;
        }  // int16 pipeline

        if (e->schedule == 0) {
            if (phases != PH_ALL) return set_err(SGM_ERR_INVALID_ARG, "phased compute needs the fused schedules");
            // -- v1 schedule: one kernel per direction, vertical-ish first, horizontal last (WTA)
            struct Dir { int rx, ry; const char *name; };
            static const Dir down[3] = {{0, 1, "path_S"}, {1, 1, "path_SE"}, {-1, 1, "path_SW"}};
            static const Dir upw[3] = {{0, -1, "path_N"}, {1, -1, "path_NE"}, {-1, -1, "path_NW"}};
            bool first = true;
            for (int k = 0; k < 3; k++) {
                if ((rc = stage_begin(e, down[k].name))) return rc;
                launch_path(g, down[k].rx, down[k].ry, first ? PATH_FIRST : PATH_ACCUM, C, S, 0, wta, st);
                KCHECK();
                first = false;
                if ((rc = stage_end(e, 1))) return rc;
            }
            if (g.mode == 1) {
                for (int k = 0; k < 3; k++) {
                    if ((rc = stage_begin(e, upw[k].name))) return rc;
                    launch_path(g, upw[k].rx, upw[k].ry, PATH_ACCUM, C, S, 0, wta, st);
                    KCHECK();
                    if ((rc = stage_end(e, 1))) return rc;
                }
            }
            if ((rc = stage_begin(e, "path_E"))) return rc;
            launch_path(g, 1, 0, PATH_ACCUM, C, S, 0, wta, st);
            KCHECK();
            if ((rc = stage_end(e, 1))) return rc;
            if ((rc = stage_begin(e, "path_W_wta"))) return rc;
            launch_path(g, -1, 0, PATH_LAST, C, S, e->keep_aggr, wta, st);
            KCHECK();
            if ((rc = stage_end(e, 1))) return rc;
        } else {
            // -- fused schedule: per pass a read-only boundary pre-pass (3 line scans) + one sweep
            // Winner-take-all: a separate pass over S (k_wta_t, one lane per pixel) after the second sweep
            // of MODE_HH and after the in-row path of MODE_SGBM for D <= 128; fused into the in-row path
            // kernel for MODE_SGBM with D > 128 (there the separate form costs a third volume of traffic:
            // 4K D=256 2.49 ms fused against 2.56 + 0.73; D=128: 2.01 against 1.42 + 0.43, 1080p 0.79 against
            // 0.44 + 0.10).  debug 2 forces the fused form everywhere, debug 2048 the separate one (A/B, cross-check).
            const bool fused_wta = !(e->debug & 2048) && (((e->debug & 2) != 0 && !rows4) ||
                                                          (g.mode == 0 && ((e->debug & 4) || g.D > 128)));
            // MODE_SGBM with the separate winner-take-all (D <= 128): the fifth path (in-row, right to left) needs
            // nothing but C, so it runs on the auxiliary stream from here on, as a FIRST pass into a volume of its
            // own (2 V of traffic instead of the 3 V of "S +="), beside the pre-pass and the sweep -- which at these
            // D are bound by instruction issue, not by HBM; k_wta_t adds the two volumes while it stages them.
            // debug 65536: the fifth path after the sweep, accumulating into S (A/B).
            const bool two_vol = g.mode == 0 && !fused_wta && g.D <= 128 && !(e->debug & 4) && !(e->debug & 65536);
            int16_t *S2 = nullptr;
            if (two_vol) {
                if ((rc = e->aggr2.ensure((size_t)g.rowsz * H * sizeof(int16_t)))) return rc;
                S2 = (int16_t *)e->aggr2.p;
            }
            // D <= 64 (small-D schedule): the OTHER in-row path (left to right) needs nothing but C either.  It used to follow
            // the element-wise vertical kernel as "S +=" on the main stream -- a chain of W1 dependent steps on the
            // critical path of a latency-bound frame; now it runs as a FIRST pass into a third volume on a stream of its
            // own, beside the per-row pre-pass and k_vert3_g, and the winner-take-all adds three volumes.
            const bool three_vol = two_vol && rows4 && !(e->debug & SGM_DBG_IN_ROW_ON_MAIN_STREAM);
            int16_t *S3 = nullptr;
            if (three_vol) {
                if ((rc = e->aggr3.ensure((size_t)g.rowsz * H * sizeof(int16_t)))) return rc;
                S3 = (int16_t *)e->aggr3.p;
            }
            // ... and so do the three directions that come from the row above: the walk along their lines (the "pre-pass" of
            // the small-D schedule) forms L_r(p, .) on its way, so each role writes it to a volume of its own and the
            // winner-take-all adds five volumes: no per-row record (3 V written, 3 V read), no element-wise kernel behind
            // the walk -- and all five directions are ONE launch (k_paths5_g: why, see there).  debug 8192: the record form
            // with the in-row paths on streams of their own (A/B; MODE_HH keeps it).
            const bool five_vol = three_vol && !(e->debug & SGM_DBG_SMALL_D_RECORD);
            int16_t *S4 = nullptr, *S5 = nullptr;
            if (five_vol) {
                if ((rc = e->aggr4.ensure((size_t)g.rowsz * H * sizeof(int16_t)))) return rc;
                if ((rc = e->aggr5.ensure((size_t)g.rowsz * H * sizeof(int16_t)))) return rc;
                S4 = (int16_t *)e->aggr4.p;
                S5 = (int16_t *)e->aggr5.p;
            }
            const size_t bnd_bytes = five_vol ? 0 : (size_t)nbands * g.W1 * 3 * g.D * 2;
            if (nbands > 1 && !five_vol) {
                if ((rc = e->bndL.ensure(bnd_bytes))) return rc;
                if (!chain) {
                    if (npass == 2 && (rc = e->bndL2.ensure(bnd_bytes))) return rc;
                    const size_t st_bytes = (size_t)2 * 3 * g.W1 * g.D * 2;  // ping-pong line state between pre-pass chunks
                    if ((rc = e->pstate.ensure(st_bytes))) return rc;
                    if (npass == 2 && (rc = e->pstate2.ensure(st_bytes))) return rc;
                }
            }
            const size_t ctl_bytes = ((size_t)(1 + nbands) * 4 + 15) & ~(size_t)15;
            if (do_pre) {
                e->plan_chain = chain;
                e->plan_R = R;
                e->plan_nbands = nbands;
            }
            if (phases != PH_ALL && !chain) return set_err(SGM_ERR_INVALID_ARG, "phased compute needs the chained schedule");
            if (chain) {
                if ((rc = e->chain_ctl.ensure(ctl_bytes))) return rc;
                if (!e->chain_err.p) {
                    if ((rc = e->chain_err.ensure(16))) return rc;
                    HIP_TRY(hipMemsetAsync(e->chain_err.p, 0, 16, st));
                }
            }
            // roles of the pre-pass: 0 = predecessor one step earlier in the sweep's x order
            // (x - xdir), 1 = same column, 2 = one step later
            // one row chunk [s0, s1) (in sweep order) of the fused three-role pre-pass; chunk index c picks the
            // ping-pong halves of the line-state buffer (kernels_path.h)
            auto prepass_chunk = [&](int xdir, int ydir, int16_t *bl, hipStream_t on, int c, int s0, int s1, bool plain) {
                const bool partial = g.D != 128 * g.NP;
                const int cpx = plain ? 0 : (g.W1 + 7) / 8;
                // (Padding the grid so that every SIMD holds the same number of waves, and halving the
                // prefetch depth, were both measured: no change -- DESIGN.md 4.4.)
                const int wpb = SGM_PREPASS_WPB;
                dim3 grid(plain ? (g.W1 + wpb - 1) / wpb : 8 * ((cpx + wpb - 1) / wpb)), block(64 * wpb);
                const size_t half = (size_t)3 * g.W1 * g.D;  // int16 elements of one state buffer
                int16_t *sbuf = (int16_t *)(ydir > 0 ? e->pstate.p : e->pstate2.p);
                const int16_t *sin = sbuf ? sbuf + (size_t)(c & 1) * half : nullptr;
                int16_t *sout = sbuf ? sbuf + (size_t)((c + 1) & 1) * half : nullptr;
#define SGM_PRE(NP_, PART_) hipLaunchKernelGGL((k_prepass3<NP_, PART_>), grid, block, 0, on, g, xdir, ydir, C, bl, R, s0, s1, sin, sout, cpx)
                // (prefetch blocks of 2 rows for the pass that runs beside the sweep -- 70 registers instead of 106 -- were
                // measured in round 2: within noise; that instantiation is gone)
                if (g.NP == 1) { if (partial) SGM_PRE(1, true); else SGM_PRE(1, false); }
                else if (g.NP == 2) { if (partial) SGM_PRE(2, true); else SGM_PRE(2, false); }
                else { if (partial) SGM_PRE(4, true); else SGM_PRE(4, false); }
#undef SGM_PRE
            };
            // roles of the pre-pass: 0 = predecessor one step earlier in the sweep's x order
            // (x - xdir), 1 = same column, 2 = one step later
            auto launch_prepass = [&](int xdir, int ydir, int16_t *bl, hipStream_t on) -> int {  // returns the launch count
                if (rows4 && !(e->debug & 16)) {  // lane-grouped lines, state stored after every row
                    // one role per wave (grid.y = 3): these frames have too few lines to fill the SIMDs with
                    // three-role waves (4K D=16: 478)
                    dim3 grid((g.W1 + 64 / GWs - 1) / (64 / GWs), 3), block(64);
                    if (GWs == 8) hipLaunchKernelGGL((k_prepass3_g<8, false>), grid, block, 0, on, g, xdir, ydir, C, bl);
                    else if (GWs == 16) hipLaunchKernelGGL((k_prepass3_g<16, false>), grid, block, 0, on, g, xdir, ydir, C, bl);
                    else if (g.D < 64) hipLaunchKernelGGL((k_prepass3_g<32, true>), grid, block, 0, on, g, xdir, ydir, C, bl);
                    else hipLaunchKernelGGL((k_prepass3_g<32, false>), grid, block, 0, on, g, xdir, ydir, C, bl);
                    return 1;
                }
                if (fused_prepass) {
                    // the three roles fused in one wave (k_prepass3); debug bit 16 selects the 3-launch variant.
                    // Row chunks of about 135 rows, one launch each, base columns grouped per XCD: two of the
                    // three reads of a C pixel hit L2 (kernels_path.h).  debug 512: one chunk, plain layout (A/B).
                    const bool plain = (e->debug & 512) != 0;
                    const int nch = plain ? 1 : (e->prepass_rows > 0 ? (H + e->prepass_rows - 1) / e->prepass_rows
                                                                          : std::max(1, (H + 67) / 135));
                    // multiples of 8 rows (two prefetch blocks): a chunk then ends in straight-line code
                    const int Hc = e->prepass_rows > 0 ? e->prepass_rows : ((H + nch - 1) / nch + 7) / 8 * 8;
                    int n = 0;
                    for (int c = 0; c < nch; c++) {
                        const int s0 = c * Hc, s1 = std::min(H, s0 + Hc);
                        if (s0 >= s1) break;
                        prepass_chunk(xdir, ydir, bl, on, c, s0, s1, plain);
                        n++;
                    }
                    return n;
                }
                // one launch of the single-direction kernel, grid.y = role
                Boundary bd{bl, R, 0};
                launch_path(g, xdir, ydir, PATH_BOUNDARY, C, S, 0, wta, on, bd);
                return 1;
            };
            // MODE_HH: the upward pre-pass only reads C, so it runs on the auxiliary stream while the
            // main stream does the downward pre-pass and sweep (memory-bound beside issue-bound work)
            const bool overlap = npass == 2 && nbands > 1 && !(e->debug & 32) && !chain;
            auto fork_prepass_up = [&]() -> int {  // aux stream: upward pre-pass, from "now" on the main stream
                int rc2;
                if (!e->aux) {
                    HIP_TRY(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(e->ev_fork, st));
                HIP_TRY(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
                if ((rc2 = stage_begin(e, "prepass_up", e->aux))) return rc2;
                const int nl = launch_prepass(-1, -1, (int16_t *)e->bndL2.p, e->aux);
                KCHECK();
                if ((rc2 = stage_end(e, nl, e->aux))) return rc2;
                HIP_TRY(hipEventRecord(e->ev_join, e->aux));
                return SGM_OK;
            };
            // debug bit 64+128: fork right after the cost stage (both pre-passes side by side) instead
            // of after the downward pre-pass (upward pre-pass beside the downward sweep)
            const bool fork_early = (e->debug & 128) != 0;
            if (two_vol && do_pre && !five_vol) {
                if (!e->aux) {
                    HIP_TRY(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(e->ev_fork, st));
                HIP_TRY(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
                if ((rc = stage_begin(e, "path_W", e->aux))) return rc;
                launch_rows_grouped(g, H, group_width(g, H), -1, PATH_FIRST, C, S2, 1, wta, e->aux);
                KCHECK();
                if ((rc = stage_end(e, 1, e->aux))) return rc;
                HIP_TRY(hipEventRecord(e->ev_join, e->aux));
                if (three_vol) {
                    if (!e->aux2) {
                        HIP_TRY(hipStreamCreateWithFlags(&e->aux2, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&e->ev_join2, hipEventDisableTiming));
                    }
                    HIP_TRY(hipStreamWaitEvent(e->aux2, e->ev_fork, 0));
                    if ((rc = stage_begin(e, "path_E", e->aux2))) return rc;
                    launch_rows_grouped(g, H, GWs, +1, PATH_FIRST, C, S3, 1, wta, e->aux2);
                    KCHECK();
                    if ((rc = stage_end(e, 1, e->aux2))) return rc;
                    HIP_TRY(hipEventRecord(e->ev_join2, e->aux2));
                }
            }
            if (overlap && fork_early && (rc = fork_prepass_up())) return rc;
            for (int pass = 0; pass < npass && do_mid; pass++) {
                const int ydir = pass == 0 ? 1 : -1, xdir = ydir;
                int16_t *bl = (int16_t *)(pass == 0 ? e->bndL.p : e->bndL2.p);
                if (nbands > 1 && !(overlap && pass == 1) && !chain && !five_vol) {
                    if ((rc = stage_begin(e, pass == 0 ? "prepass_dn" : "prepass_up"))) return rc;
                    const int nl = launch_prepass(xdir, ydir, bl, st);
                    KCHECK();
                    if ((rc = stage_end(e, nl))) return rc;
                }
                if (overlap && pass == 0 && !fork_early && (rc = fork_prepass_up())) return rc;
                if (overlap && pass == 1) {
                    HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));
                    stage_break(e);  // (the wait is not part of the next stage)
                }
                // winner-take-all: fused into the last path kernel (debug 2), or -- default -- a
                // separate pass over S with one lane per pixel (k_wta_t)
                const bool last = pass == npass - 1 && g.mode == 1 && fused_wta && !rows4;
                SweepArgs a{ydir, xdir, R, C, S, (const int16_t *)bl, wta, e->keep_aggr, e->debug, nullptr, nullptr, nullptr, nbands};
                ChainFrames fr;
                if (chain) {
                    // one record serves both passes (they follow each other on the stream)
                    fr.nf = 1;
                    fr.C[0] = C;
                    fr.S[0] = S;
                    fr.bnd[0] = (int16_t *)e->bndL.p;
                    fr.hr[0] = g.hr;
                    a.ctl = (uint32_t *)e->chain_ctl.p;
                    a.err = (uint32_t *)e->chain_err.p;
                    HIP_TRY(hipMemsetAsync(e->chain_ctl.p, 0, ctl_bytes, st));
                    stage_break(e);
                }
                if ((rc = stage_begin(e, five_vol ? "paths5" : chain ? (pass == 0 ? "chain_dn" : "chain_up")
                                               : (pass == 0 ? "sweep_dn" : (fused_wta ? "sweep_up_wta" : "sweep_up"))))) return rc;
                // per-row state written by the grouped pre-pass: role-major; by the single-direction kernel (debug 16): band layout
                const int rmaj = (e->debug & 16) ? 0 : 1;
                if (five_vol) {
                    // D <= 64, MODE_SGBM: all five directions in one launch, one volume each (k_paths5_g)
                    const int G = 64 / GWs, nr = (H + G - 1) / G, nl = (g.W1 + G - 1) / G;
                    // (Measured beside it: rows and lines as two launches one after the other, in either order -- 4K D=16 0.60 /
                    // 0.59 ms against 0.57 ms, 720p D=64 0.21 / 0.22 against 0.19; occupancy capped at one or two waves per SIMD
                    // through an LDS allocation -- 0.66 against 0.65 ms, 0.23 / 0.27 against 0.20.  Neither the order nor the
                    // number of waves per SIMD matters: the stage moves 6 ... 10 V in 32-byte pieces per row and is bound by
                    // the memory side)
                    dim3 grid(2 * nr + 3 * nl), block(64);
                    if (GWs == 8) hipLaunchKernelGGL((k_paths5_g<8, false>), grid, block, 0, st, g, xdir, ydir, C, S, S4, S5, S2, S3, nr);
                    else if (GWs == 16) hipLaunchKernelGGL((k_paths5_g<16, false>), grid, block, 0, st, g, xdir, ydir, C, S, S4, S5, S2, S3, nr);
                    else if (g.D < 64) hipLaunchKernelGGL((k_paths5_g<32, true>), grid, block, 0, st, g, xdir, ydir, C, S, S4, S5, S2, S3, nr);
                    else hipLaunchKernelGGL((k_paths5_g<32, false>), grid, block, 0, st, g, xdir, ydir, C, S, S4, S5, S2, S3, nr);
                } else if (rows4) {
                    // D <= 64, band height 1: the three directions from the previous row are element-wise given the
                    // pre-pass state of every row (k_vert3_g, one streaming pass over all pixels); only the in-row
                    // direction is a recurrence (k_rows_g, S +=)
                    dim3 grid((g.W1 + 255) / 256, H), block(256);
                    const int16_t *bq = (const int16_t *)bl;
#define SGM_VERT(GW_, PART_)                                                                                     \
    do {                                                                                                         \
        if (pass == 0) hipLaunchKernelGGL((k_vert3_g<GW_, PATH_FIRST, PART_>), grid, block, 0, st, g, xdir, ydir, C, S, bq, rmaj); \
        else hipLaunchKernelGGL((k_vert3_g<GW_, PATH_ACCUM, PART_>), grid, block, 0, st, g, xdir, ydir, C, S, bq, rmaj); \
    } while (0)
                    if (GWs == 8) SGM_VERT(8, false);
                    else if (GWs == 16) SGM_VERT(16, false);
                    else if (g.D < 64) SGM_VERT(32, true);
                    else SGM_VERT(32, false);
#undef SGM_VERT
                    if (!three_vol) launch_rows_grouped(g, H, GWs, xdir, PATH_ACCUM, C, S, 1, wta, st);
                } else if (chain) {
                    if ((rc = launch_chain(g, a, fr, pass == 0 ? SWEEP_FIRST : SWEEP_ACCUM, chain_window(g, R, nbands, 1, e->chain_wgs), st))) return rc;
                } else if ((rc = launch_sweep(g, a, pass == 0 ? SWEEP_FIRST : (last ? SWEEP_LAST : SWEEP_ACCUM), nbands, st))) {
                    return rc;
                }
                KCHECK();
                if ((rc = stage_end(e, 1))) return rc;
            }
            if (!do_post) return SGM_OK;
            if (two_vol && !five_vol) {
                HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));
                if (three_vol) HIP_TRY(hipStreamWaitEvent(st, e->ev_join2, 0));
                stage_break(e);
            }
            if (g.mode == 0 && !two_vol) {
                if ((rc = stage_begin(e, fused_wta ? "path_W_wta" : "path_W"))) return rc;
                const int GW = (e->debug & 4) ? 64 : group_width(g, H);  // debug 4: no lane groups (A/B)
                const int pm = fused_wta ? PATH_LAST : PATH_ACCUM;
                if (GW == 64 && (e->debug & 4)) launch_path(g, -1, 0, pm, C, S, e->keep_aggr, wta, st);  // A/B: the general line kernel
                else launch_rows_grouped(g, H, GW, -1, pm, C, S, e->keep_aggr, wta, st);
                KCHECK();
                if ((rc = stage_end(e, 1))) return rc;
            }
            if (!fused_wta) {
                if ((rc = stage_begin(e, "wta"))) return rc;
                const int64_t npix = (int64_t)H * g.W1;
                const size_t lds = (size_t)64 * wta_t_stride(g.D);
                // persistent blocks: LDS (64 padded rows) allows four waves per CU; each loops over its share
                dim3 grid((unsigned)std::min<int64_t>((npix + 63) / 64, 4 * 256)), block(64);
                int lgc = -1;  // log2(D / 8) when D is a power of two
                for (int q = 1; q <= 6; q++)
                    if (g.D == (8 << q)) lgc = q;
#define SGM_WTA(POSW_, LG_)                                                                                            \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            HIP_TRY(hipFuncSetAttribute((const void *)k_wta_t<POSW_, LG_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                    \
        hipLaunchKernelGGL((k_wta_t<POSW_, LG_>), grid, block, lds, st, g, (const int16_t *)S, wta, npix,              \
                           (const int16_t *)nullptr, (const int16_t *)nullptr, (const int16_t *)nullptr,              \
                           (const int16_t *)nullptr);                                                                  \
    } while (0)
#define SGM_WTA2(POSW_, LG_)                                                                                           \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            HIP_TRY(hipFuncSetAttribute((const void *)k_wta_t<POSW_, LG_, 2>,                                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL((k_wta_t<POSW_, LG_, 2>), grid, block, lds, st, g, (const int16_t *)S, wta, npix,           \
                           (const int16_t *)S2, (const int16_t *)nullptr, (const int16_t *)nullptr,                    \
                           (const int16_t *)nullptr);                                                                  \
    } while (0)
#define SGM_WTA3(POSW_, LG_)                                                                                           \
    do {                                                                                                               \
        if (five_vol)                                                                                                  \
            hipLaunchKernelGGL((k_wta_t<POSW_, LG_, 5>), grid, block, lds, st, g, (const int16_t *)S, wta, npix,       \
                               (const int16_t *)S2, (const int16_t *)S3, (const int16_t *)S4, (const int16_t *)S5);   \
        else                                                                                                           \
            hipLaunchKernelGGL((k_wta_t<POSW_, LG_, 3>), grid, block, lds, st, g, (const int16_t *)S, wta, npix,       \
                               (const int16_t *)S2, (const int16_t *)S3, (const int16_t *)nullptr,                     \
                               (const int16_t *)nullptr);                                                              \
    } while (0)
#define SGM_WTA3_LG(POSW_)                      \
    do {                                        \
        switch (lgc) {                          \
        case 1: SGM_WTA3(POSW_, 1); break;      \
        case 2: SGM_WTA3(POSW_, 2); break;      \
        case 3: SGM_WTA3(POSW_, 3); break;      \
        default: SGM_WTA3(POSW_, -1); break;    \
        }                                       \
    } while (0)
#define SGM_WTA2_LG(POSW_)                      \
    do {                                        \
        switch (lgc) {                          \
        case 1: SGM_WTA2(POSW_, 1); break;      \
        case 2: SGM_WTA2(POSW_, 2); break;      \
        case 3: SGM_WTA2(POSW_, 3); break;      \
        case 4: SGM_WTA2(POSW_, 4); break;      \
        default: SGM_WTA2(POSW_, -1); break;    \
        }                                       \
    } while (0)
#define SGM_WTA_LG(POSW_)                       \
    do {                                        \
        switch (lgc) {                          \
        case 1: SGM_WTA(POSW_, 1); break;       \
        case 2: SGM_WTA(POSW_, 2); break;       \
        case 3: SGM_WTA(POSW_, 3); break;       \
        case 4: SGM_WTA(POSW_, 4); break;       \
        case 5: SGM_WTA(POSW_, 5); break;       \
        case 6: SGM_WTA(POSW_, 6); break;       \
        default: SGM_WTA(POSW_, -1); break;     \
        }                                       \
    } while (0)
                if (two_vol) {
                    if (three_vol) {
                        if (g.uniq < 100) SGM_WTA3_LG(true);
                        else SGM_WTA3_LG(false);
                    } else if (g.uniq < 100) SGM_WTA2_LG(true);
                    else SGM_WTA2_LG(false);
                    if (e->keep_aggr) {  // the volume a caller inspects is the whole sum
                        const int64_t n8 = (int64_t)g.rowsz * H / 8;  // rowsz = W1 * D is a multiple of 16
                        hipLaunchKernelGGL(k_add_sat, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, S, (const int16_t *)S2, n8);
                        if (three_vol) hipLaunchKernelGGL(k_add_sat, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, S, (const int16_t *)S3, n8);
                        if (five_vol) {
                            hipLaunchKernelGGL(k_add_sat, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, S, (const int16_t *)S4, n8);
                            hipLaunchKernelGGL(k_add_sat, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, S, (const int16_t *)S5, n8);
                        }
                    }
                } else if (g.uniq < 100) SGM_WTA_LG(true);
                else SGM_WTA_LG(false);
#undef SGM_WTA3_LG
#undef SGM_WTA3
#undef SGM_WTA2_LG
#undef SGM_WTA2
#undef SGM_WTA_LG
#undef SGM_WTA
                KCHECK();
                if ((rc = stage_end(e, 1))) return rc;
            }
        }

        // -- right view, sub-pixel, LR check
        if ((rc = stage_begin(e, "select_lr"))) return rc;
        {
            const size_t lds = (size_t)W * 4;
            if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_select, dim3(H), dim3(256), lds, st, g, (const uint2 *)wta, raw);
            KCHECK();
        }
        if ((rc = stage_end(e, 1))) return rc;
    }

    // -- 3x3 median
    if ((rc = stage_begin(e, "median3"))) return rc;
    {
        dim3 grid((W + 255) / 256, H), block(256);
        // the result goes to the median tap AND to the output buffer (the speckle filter works in place there)
        hipLaunchKernelGGL(k_median3, grid, block, 0, st, (const int16_t *)raw, med, d_disp, H, W);
        KCHECK();
    }
    if ((rc = stage_end(e, 1))) return rc;

    // -- speckle filter on the output
    if (e->params.speckleRange >= 0 && e->params.speckleWindowSize > 0) {  // upstream's condition for filterSpeckles
        if ((rc = stage_begin(e, "speckle"))) return rc;
        if ((rc = run_speckles(e, d_disp, H, W, (e->params.minDisparity - 1) * 16, e->params.speckleWindowSize,
                               16 * e->params.speckleRange)))
            return rc;
        if ((rc = stage_end(e, 4))) return rc;
    }
    return SGM_OK;
}

static int run_to_float(sgm_engine *e, const int16_t *d_disp, int64_t n, float *d_out)
{
    hipLaunchKernelGGL(k_disp_to_float, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, d_disp, d_out, n);
    KCHECK();
    return SGM_OK;
}

static int run_reproject(sgm_engine *e, const float *d_disp, int H, int W, const double Q[16], int handle_missing,
                         float *d_xyz)
{
    if (!Q) return set_err(SGM_ERR_INVALID_ARG, "Q is null");
    QMat q;
    memcpy(q.q, Q, sizeof(q.q));
    uint32_t *mk = nullptr;
    if (handle_missing) {
        int rc = e->minkey.ensure(4);
        if (rc) return rc;
        mk = (uint32_t *)e->minkey.p;
        HIP_TRY(hipMemsetAsync(mk, 0xff, 4, e->stream));
        const int64_t n = (int64_t)H * W;
        const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(k_min_f32, dim3(nb), dim3(256), 0, e->stream, d_disp, n, mk);
    }
    dim3 grid((W + 255) / 256, H), block(256);
    hipLaunchKernelGGL(k_reproject, grid, block, 0, e->stream, d_disp, H, W, q, (const uint32_t *)mk, d_xyz);
    KCHECK();
    return SGM_OK;
}

// float scaling + reprojection of one pair on its engine's stream (shared by the single-pair and the batch entry)
static int run_float_xyz(sgm_engine *e, const int16_t *di, int H, int W, const double Q[16], void *d_disp_f32, void *d_xyz_f32)
{
    const int64_t n = (int64_t)H * W;
    int rc;
    if (!d_disp_f32 && !d_xyz_f32) return SGM_OK;
    if (d_xyz_f32) {
        if (!Q) return set_err(SGM_ERR_INVALID_ARG, "Q is null");
        QMat q;
        memcpy(q.q, Q, sizeof(q.q));
        if ((rc = stage_begin(e, "float_xyz"))) return rc;
        hipLaunchKernelGGL(k_float_xyz, dim3((W + 255) / 256, H), dim3(256), 0, e->stream, di, H, W, q,
                           (float *)d_disp_f32, (float *)d_xyz_f32);
        KCHECK();
        return stage_end(e, 1);
    }
    if ((rc = stage_begin(e, "to_float"))) return rc;
    if ((rc = run_to_float(e, di, n, (float *)d_disp_f32))) return rc;
    return stage_end(e, 1);
}

// ---- exported C ABI -------------------------------------------------------------------------------
extern "C" {

int sgm_abi_version(void) { return SGM_ABI_VERSION; }

const char *sgm_last_error(void) { return g_err; }

int sgm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sgm_geometry(const sgm_params *params, int W, int *minX1, int *W1)
{
    if (!params) return set_err(SGM_ERR_INVALID_ARG, "params is null");
    Geom g;
    int rc = normalise(params, 1, W, &g);
    if (rc) return rc;
    if (minX1) *minX1 = g.minX1;
    if (W1) *W1 = g.W1;
    return SGM_OK;
}

int sgm_create(const sgm_params *params, int device_id, void *stream, sgm_engine **out)
{
    if (!params || !out) return set_err(SGM_ERR_INVALID_ARG, "params/out is null");
    *out = nullptr;
    Geom g;
    int rc = normalise(params, 1, 64, &g);  // parameter validation only
    if (rc) return rc;
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess || n <= 0)
        return set_err(SGM_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                       he == hipSuccess ? "device count 0" : hipGetErrorString(he));
    if (device_id < 0 || device_id >= n) return set_err(SGM_ERR_NO_DEVICE, "device_id %d out of range [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    sgm_engine *e = new (std::nothrow) sgm_engine();
    if (!e) return set_err(SGM_ERR_NOMEM, "out of host memory");
    e->params = *params;
    e->device = device_id;
    if (stream) {
        e->stream = (hipStream_t)stream;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete e;
            return set_err(SGM_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se));
        }
        e->own_stream = true;
    }
    *out = e;
    return SGM_OK;
}

// every device buffer of an engine (the engine stays usable: buffers come back on the next call that needs them)
static void release_buffers(sgm_engine *e)
{
    DevBuf *bufs[] = {&e->in_left, &e->in_right, &e->lrec, &e->rplanes, &e->hsum, &e->cost, &e->aggr, &e->aggr2, &e->rmap1, &e->rmap2, &e->rsrc, &e->rdst, &e->wta, &e->bndL, &e->bndL2, &e->pstate, &e->pstate2,
                      &e->disp_raw, &e->disp_med, &e->disp_out, &e->label, &e->csize, &e->rlen, &e->f32, &e->xyz, &e->mask,
                      &e->minkey, &e->ccount, &e->cpts, &e->crgb, &e->crgb_in, &e->headroom, &e->chain_ctl, &e->chain_err, &e->aggr3, &e->aggr4, &e->aggr5};
    for (DevBuf *b : bufs) (void)b->release();
    for (auto &slot : e->io)
        for (DevBuf &b : slot) (void)b.release();
    e->g.hr = nullptr;
}

void sgm_destroy(sgm_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->copy_in) {
        (void)hipStreamSynchronize(e->copy_in);
        (void)hipStreamDestroy(e->copy_in);
    }
    if (e->copy_out) {
        (void)hipStreamSynchronize(e->copy_out);
        (void)hipStreamDestroy(e->copy_out);
    }
    release_buffers(e);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    for (auto *evs : {e->ev_io_in, e->ev_io_used, e->ev_io_out, e->ev_io_dl})
        for (int k = 0; k < 2; k++)
            if (evs[k]) (void)hipEventDestroy(evs[k]);
    for (auto &slot : e->pin_io)
        for (HostBuf &b : slot) b.release();
    if (e->aux2) {
        (void)hipStreamSynchronize(e->aux2);
        (void)hipEventDestroy(e->ev_join2);
        (void)hipStreamDestroy(e->aux2);
    }
    if (e->aux) {
        (void)hipStreamSynchronize(e->aux);
        (void)hipEventDestroy(e->ev_fork);
        (void)hipEventDestroy(e->ev_join);
        (void)hipStreamDestroy(e->aux);
    }
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    e->pin_left.release();
    e->pin_right.release();
    e->pin_disp.release();
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    if (e->peer) sgm_destroy(e->peer);
    if (e->peer2) sgm_destroy(e->peer2);
    for (sgm_engine *q : e->group) sgm_destroy(q);
    if (e->ev_group) (void)hipEventDestroy(e->ev_group);
    delete e;
}

int sgm_set_option(sgm_engine *e, int option, int value)
{
    if (!e) return set_err(SGM_ERR_INVALID_ARG, "engine is null");
    if (option == SGM_OPT_KEEP_AGGR) e->keep_aggr = value ? 1 : 0;
    else if (option == SGM_OPT_PROFILE) e->profile = value ? 1 : 0;
    else if (option == SGM_OPT_SCHEDULE) e->schedule = std::max(0, std::min(value, 2));
    else if (option == SGM_OPT_CHAIN_WGS) e->chain_wgs = std::max(0, value);
    else if (option == SGM_OPT_SWEEP_ROWS) e->sweep_rows = std::max(0, value);
    else if (option == SGM_OPT_DEBUG) {
        // csrc/sgm_debug.h; bit 64 (the sweep's loader skips its loads: timing only, results WRONG) must be asked for twice
        if ((value & SGM_DBG_SKIP_BOUNDARY_LOADS) && !(getenv("SGM_ALLOW_WRONG_RESULTS") && atoi(getenv("SGM_ALLOW_WRONG_RESULTS")) == 1))
            return set_err(SGM_ERR_INVALID_ARG, "debug bit 64 makes results wrong (timing experiments only): set SGM_ALLOW_WRONG_RESULTS=1 to use it");
        e->debug = value;
    }
    else if (option == SGM_OPT_PREPASS_ROWS) e->prepass_rows = std::max(0, value);
    else if (option == SGM_OPT_GROUP_MAX) e->group_max = std::max(0, std::min(value, CHAIN_MAX_FRAMES));
    else return set_err(SGM_ERR_INVALID_ARG, "unknown option %d", option);
    return SGM_OK;
}

// Chained sweeps bound every wait for another workgroup (kernels_sweep.h: ChainWait); a wait that gave up leaves a
// flag beside the control words and wrong results in whatever was computed since the last check.  The flag is looked at
// wherever the host synchronises with the engine's stream anyway, and by sgm_check for callers that order the engine's
// stream with events of their own (dist.IngestPipeline, bench.py); once reported it is cleared, so that the engine is
// usable again (the next launch zeroes its control words as every launch does).
static int check_chain(sgm_engine *e)
{
    if (!e->chain_err.p) return SGM_OK;
    uint32_t f = 0;
    HIP_TRY(hipMemcpy(&f, e->chain_err.p, 4, hipMemcpyDeviceToHost));
    if (f) {
        HIP_TRY(hipMemset(e->chain_err.p, 0, 4));
        return set_err(SGM_ERR_HIP, "chained sweep: a workgroup gave up waiting for the band above it (results since the last check are invalid)");
    }
    return SGM_OK;
}

int sgm_synchronize(sgm_engine *e)
{
    if (!e) return set_err(SGM_ERR_INVALID_ARG, "engine is null");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return check_chain(e);
}

int sgm_check(sgm_engine *e)
{
    if (!e) return set_err(SGM_ERR_INVALID_ARG, "engine is null");
    HIP_TRY(hipSetDevice(e->device));
    return check_chain(e);
}

// gives back what the engine can build again: the internal engines of sgm_pipeline_batch_device / sgm_compute_batch (a
// group of 4K D=256 pairs holds about 9 GB per pair) and the page-locked staging buffers
int sgm_trim(sgm_engine *e)
{
    if (!e) return set_err(SGM_ERR_INVALID_ARG, "engine is null");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (sgm_engine *q : e->group) sgm_destroy(q);
    e->group.clear();
    e->last_group = 0;
    if (e->peer) sgm_destroy(e->peer);
    if (e->peer2) sgm_destroy(e->peer2);
    e->peer = e->peer2 = nullptr;
    e->pin_left.release();
    e->pin_right.release();
    e->pin_disp.release();
    for (auto &slot : e->pin_io)
        for (HostBuf &b : slot) b.release();
    return check_chain(e);
}

int sgm_compute_device(sgm_engine *e, const void *d_left, const void *d_right, int H, int W, int64_t stride_bytes,
                       void *d_disp_i16)
{
    return run_compute(e, (const uint8_t *)d_left, (const uint8_t *)d_right, H, W, stride_bytes, (int16_t *)d_disp_i16);
}

int sgm_disp_to_float_device(sgm_engine *e, const void *d_disp_i16, int64_t n, void *d_out_f32)
{
    if (!e || !d_disp_i16 || !d_out_f32 || n <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    return run_to_float(e, (const int16_t *)d_disp_i16, n, (float *)d_out_f32);
}

int sgm_reproject_device(sgm_engine *e, const void *d_disp_f32, int H, int W, const double Q[16], int handle_missing,
                         void *d_xyz_f32)
{
    if (!e || !d_disp_f32 || !d_xyz_f32 || H <= 0 || W <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    return run_reproject(e, (const float *)d_disp_f32, H, W, Q, handle_missing, (float *)d_xyz_f32);
}

int sgm_valid_mask_device(sgm_engine *e, const void *d_xyz, const void *d_disp_f32, int64_t n, void *d_mask_u8)
{
    if (!e || !d_xyz || !d_disp_f32 || !d_mask_u8 || n <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_valid_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, (const float *)d_xyz,
                       (const float *)d_disp_f32, n, (uint8_t *)d_mask_u8);
    KCHECK();
    return SGM_OK;
}

// ---- rectification in front of the path (gui.py:160-164) ----------------------------------------

// (P[:, :3] * R)^-1 in double, the closed form cv::invert uses for 3x3 (lapack.cpp)
static bool rectify_inverse(const double K[9], const double *R, const double *P, int pcols, double ir[9])
{
    double A[9], Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, M[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) A[r * 3 + c] = P ? P[r * pcols + c] : K[r * 3 + c];
    if (R) memcpy(Rm, R, sizeof(Rm));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int q = 0; q < 3; q++) s += A[r * 3 + q] * Rm[q * 3 + c];
            M[r * 3 + c] = s;
        }
    const double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7], m22 = M[8];
    double d = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
    if (d == 0.) return false;
    d = 1. / d;
    ir[0] = (m11 * m22 - m12 * m21) * d;
    ir[1] = (m02 * m21 - m01 * m22) * d;
    ir[2] = (m01 * m12 - m02 * m11) * d;
    ir[3] = (m12 * m20 - m10 * m22) * d;
    ir[4] = (m00 * m22 - m02 * m20) * d;
    ir[5] = (m02 * m10 - m00 * m12) * d;
    ir[6] = (m10 * m21 - m11 * m20) * d;
    ir[7] = (m01 * m20 - m00 * m21) * d;
    ir[8] = (m00 * m11 - m01 * m10) * d;
    return true;
}

int sgm_init_undistort_rectify_map_device(sgm_engine *e, const double K[9], const double *dist, int ndist,
                                          const double *R, const double *P, int pcols, int W, int H,
                                          void *d_map1_f32, void *d_map2_f32)
{
    if (!e || !K || !d_map1_f32 || !d_map2_f32 || W <= 0 || H <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (P && pcols != 3 && pcols != 4) return set_err(SGM_ERR_INVALID_ARG, "newCameraMatrix must be 3x3 or 3x4");
    RectifyArgs a;
    memset(&a, 0, sizeof(a));
    if (dist) {
        if (ndist == 14) return set_err(SGM_ERR_UNSUPPORTED, "tilted-sensor distortion terms are not supported");
        if (ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12)
            return set_err(SGM_ERR_INVALID_ARG, "distCoeffs must hold 4, 5, 8 or 12 values");
        for (int i = 0; i < ndist; i++) a.k[i] = dist[i];
    }
    if (!rectify_inverse(K, R, P, pcols, a.ir)) return set_err(SGM_ERR_INVALID_ARG, "newCameraMatrix * R is singular");
    a.fx = K[0];
    a.fy = K[4];
    a.u0 = K[2];
    a.v0 = K[5];
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_rectify_map, dim3((H + 63) / 64), dim3(64), 0, e->stream, a, W, H, (float *)d_map1_f32, (float *)d_map2_f32);
    KCHECK();
    return SGM_OK;
}

int sgm_init_undistort_rectify_map(sgm_engine *e, const double K[9], const double *dist, int ndist, const double *R,
                                   const double *P, int pcols, int W, int H, float *map1_out, float *map2_out)
{
    if (!e || !map1_out || !map2_out || W <= 0 || H <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    const size_t bytes = (size_t)W * H * 4;
    int rc;
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = e->rmap1.ensure(bytes)) || (rc = e->rmap2.ensure(bytes))) return rc;
    if ((rc = sgm_init_undistort_rectify_map_device(e, K, dist, ndist, R, P, pcols, W, H, e->rmap1.p, e->rmap2.p))) return rc;
    HIP_TRY(hipMemcpyAsync(map1_out, e->rmap1.p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(map2_out, e->rmap2.p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_remap_linear_u8_device(sgm_engine *e, const void *d_src, int sH, int sW, int64_t sstride, int cn,
                               const void *d_map1_f32, const void *d_map2_f32, int dH, int dW, void *d_dst,
                               int64_t dstride)
{
    if (!e || !d_src || !d_map1_f32 || !d_map2_f32 || !d_dst || sH <= 0 || sW <= 0 || dH <= 0 || dW <= 0)
        return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (cn < 1 || cn > 4) return set_err(SGM_ERR_UNSUPPORTED, "remap supports 1..4 interleaved 8-bit channels, got %d", cn);
    if (sstride < (int64_t)sW * cn || dstride < (int64_t)dW * cn) return set_err(SGM_ERR_INVALID_ARG, "stride smaller than a row");
    if (sH > 32767 || sW > 32767) return set_err(SGM_ERR_UNSUPPORTED, "source larger than 32767 pixels (upstream's int16 coordinates)");
    HIP_TRY(hipSetDevice(e->device));
    dim3 grid((dW + 255) / 256, dH), block(256);
    const uint8_t *s = (const uint8_t *)d_src;
    const float *m1 = (const float *)d_map1_f32, *m2 = (const float *)d_map2_f32;
    uint8_t *d = (uint8_t *)d_dst;
    switch (cn) {
    case 1: hipLaunchKernelGGL(k_remap_linear<1>, grid, block, 0, e->stream, s, sH, sW, sstride, m1, m2, dH, dW, d, dstride); break;
    case 2: hipLaunchKernelGGL(k_remap_linear<2>, grid, block, 0, e->stream, s, sH, sW, sstride, m1, m2, dH, dW, d, dstride); break;
    case 3: hipLaunchKernelGGL(k_remap_linear<3>, grid, block, 0, e->stream, s, sH, sW, sstride, m1, m2, dH, dW, d, dstride); break;
    default: hipLaunchKernelGGL(k_remap_linear<4>, grid, block, 0, e->stream, s, sH, sW, sstride, m1, m2, dH, dW, d, dstride); break;
    }
    KCHECK();
    return SGM_OK;
}

int sgm_remap_linear_u8(sgm_engine *e, const uint8_t *src, int sH, int sW, int64_t sstride, int cn, const float *map1,
                        const float *map2, int dH, int dW, uint8_t *dst)
{
    if (!e || !src || !map1 || !map2 || !dst || sH <= 0 || sW <= 0 || dH <= 0 || dW <= 0)
        return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (cn < 1 || cn > 4) return set_err(SGM_ERR_UNSUPPORTED, "remap supports 1..4 interleaved 8-bit channels, got %d", cn);
    if (sstride < (int64_t)sW * cn) return set_err(SGM_ERR_INVALID_ARG, "stride smaller than a row");
    HIP_TRY(hipSetDevice(e->device));
    const size_t sbytes = (size_t)sH * sW * cn, mbytes = (size_t)dH * dW * 4, dbytes = (size_t)dH * dW * cn;
    int rc;
    if ((rc = e->rsrc.ensure(sbytes)) || (rc = e->rmap1.ensure(mbytes)) || (rc = e->rmap2.ensure(mbytes)) || (rc = e->rdst.ensure(dbytes)))
        return rc;
    HIP_TRY(hipMemcpy2DAsync(e->rsrc.p, (size_t)sW * cn, src, (size_t)sstride, (size_t)sW * cn, sH, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->rmap1.p, map1, mbytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->rmap2.p, map2, mbytes, hipMemcpyHostToDevice, e->stream));
    if ((rc = sgm_remap_linear_u8_device(e, e->rsrc.p, sH, sW, (int64_t)sW * cn, cn, e->rmap1.p, e->rmap2.p, dH, dW, e->rdst.p, (int64_t)dW * cn)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dst, e->rdst.p, dbytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_pipeline_device(sgm_engine *e, const void *d_left, const void *d_right, int H, int W, int64_t stride_bytes,
                        const double Q[16], void *d_disp_i16, void *d_disp_f32, void *d_xyz_f32)
{
    if (!e) return set_err(SGM_ERR_INVALID_ARG, "engine is null");
    const int64_t n = (int64_t)H * W;
    int rc;
    int16_t *di = (int16_t *)d_disp_i16;
    if (!di) {
        if ((rc = e->disp_out.ensure((size_t)n * 2))) return rc;
        di = (int16_t *)e->disp_out.p;
    }
    if ((rc = run_compute(e, (const uint8_t *)d_left, (const uint8_t *)d_right, H, W, stride_bytes, di))) return rc;
    // float scaling + reprojection in one launch (the float map is stored only if asked for)
    return run_float_xyz(e, di, H, W, Q, d_disp_f32, d_xyz_f32);
}

// ---- N pairs, throughput mode --------------------------------------------------------------------------------
// With the chained schedule (SGM_OPT_SCHEDULE 2 on `e`) and a configuration the fused sweeps cover, the pairs go through
// the frame in lockstep of its phases: cost stage of every pair (each on the stream of one of up to CHAIN_MAX_FRAMES
// internal engines), then ONE chained sweep launch per pass over all pairs of the group on `e`'s stream, then the
// winner-take-all and the epilogue of every pair.  One frame's chain of bands keeps only about 50 workgroups busy; a
// group of 6 or more fills the GPU, and no boundary pre-pass runs at all.

// Everything one pair of a chained group needs on its engine, allocated BEFORE anything is enqueued (so that running out
// of memory costs nothing but a smaller group)
static int prepare_pair_buffers(sgm_engine *q, int H, int W, const Plan &pl)
{
    int rc = ensure_buffers(q, H, W);
    if (rc) return rc;
    const Geom &g = q->g;
    const size_t npx = (size_t)H * W;
    if (pl.nbands > 1 && (rc = q->bndL.ensure((size_t)pl.nbands * g.W1 * 3 * g.D * 2))) return rc;
    if (q->params.speckleRange >= 0 && q->params.speckleWindowSize > 0 &&
        ((rc = q->label.ensure(npx * 4)) || (rc = q->csize.ensure(npx * 4)) || (rc = q->rlen.ensure(npx * 4))))
        return rc;
    const bool two_vol = g.mode == 0 && g.D <= 128 && !(q->debug & (4 | 2 | 65536));   // (run_compute: the fifth path's own volume)
    if (two_vol && (rc = q->aggr2.ensure((size_t)g.rowsz * H * sizeof(int16_t)))) return rc;
    return SGM_OK;
}

// The engines of a chained group for up to `want` pairs: e itself + internal engines, created, configured like e and
// sized for the shape.  Fewer than `want` when device memory runs out (or would drop below a reserve of 4 GiB / 5 %:
// the caller and the runtime need room too) -- a batch then simply takes more groups.  *n_out >= 1.
static int prepare_group(sgm_engine *e, int want, int H, int W, const Plan &pl, int *n_out, size_t extra_per_pair = 0)
{
    int rc;
    *n_out = 0;
    if (e->group_max > 0) want = std::min(want, e->group_max);
    want = std::max(1, std::min(want, CHAIN_MAX_FRAMES));
    if ((rc = prepare_pair_buffers(e, H, W, pl))) return rc;
    *n_out = 1;
    if (want > 1 && !e->ev_group) HIP_TRY(hipEventCreateWithFlags(&e->ev_group, hipEventDisableTiming));
    bool retried = false;
    for (int k = 1; k < want; k++) {
        const bool is_new = (int)e->group.size() < k;
        if (is_new) {
            sgm_engine *q = nullptr;
            if ((rc = sgm_create(&e->params, e->device, nullptr, &q))) return rc;
            e->group.push_back(q);
        }
        sgm_engine *q = e->group[k - 1];
        q->keep_aggr = 0;
        q->profile = 0;
        q->schedule = e->schedule;
        q->sweep_rows = e->sweep_rows;
        q->debug = e->debug;
        q->chain_wgs = e->chain_wgs;
        if (!q->ev_done) HIP_TRY(hipEventCreateWithFlags(&q->ev_done, hipEventDisableTiming));
        const bool sized = q->H == H && q->W == W && q->cost.p;     // (ran this shape before: nothing to allocate)
        rc = prepare_pair_buffers(q, H, W, pl);
        size_t fr = 0, tot = 0;
        // (the reserve: 4 GiB or 5 % -- every stream, event pool and first launch of a kernel costs the runtime device memory
        //  too, and "out of memory" from a kernel launch cannot be recovered from -- plus what the caller of this function
        //  is about to allocate per pair: the host entry's transfer slots)
        if (!rc && !sized && hipMemGetInfo(&fr, &tot) == hipSuccess &&
            fr < std::max<size_t>((size_t)4 << 30, tot / 20) + extra_per_pair * (size_t)(k + 1))
            rc = SGM_ERR_NOMEM;
        if (rc == SGM_ERR_NOMEM) {
            release_buffers(q);   // (a half-sized engine would only hold memory the smaller group could use)
            q->H = q->W = 0;
            // engines behind this one may still hold the buffers of an earlier, larger batch or another shape: give those
            // back once and try this engine again before settling for a smaller group
            bool freed = false;
            for (size_t j = (size_t)k; j < e->group.size(); j++)
                if (e->group[j]->cost.p || e->group[j]->aggr.p) {
                    release_buffers(e->group[j]);
                    e->group[j]->H = e->group[j]->W = 0;
                    freed = true;
                }
            if (freed && !retried) {
                retried = true;
                k--;
                continue;
            }
            break;
        }
        if (rc) return rc;
        *n_out = k + 1;
    }
    return SGM_OK;
}

// every stream a batch call may have work on is drained before an error is returned, and the engines get their
// per-call flags back
struct BatchGuard {
    sgm_engine *e;
    bool ok = false;
    ~BatchGuard()
    {
        e->hr_accumulate = false;
        for (sgm_engine *q : e->group) q->hr_accumulate = false;
        if (ok) return;
        (void)hipStreamSynchronize(e->stream);
        for (sgm_engine *q : e->group) (void)hipStreamSynchronize(q->stream);
        if (e->copy_in) (void)hipStreamSynchronize(e->copy_in);
        if (e->copy_out) (void)hipStreamSynchronize(e->copy_out);
    }
};

// One group (n >= 2 pairs on eng[0 .. n-1], eng[0] = e) through cost stages, joint sweeps, epilogues.  The host entry
// passes three event arrays (all null for resident pairs): in_ready[k] -- pair k's cost stage waits for it (its images have
// arrived); in_used[k] -- recorded when pair k's images have been read for the last time (k_features); out_done[k] -- recorded
// behind pair k's last kernel.
static int run_group(sgm_engine *e, sgm_engine *const *eng, int n, const Plan &pl, const void *const *d_left, const void *const *d_right,
                     int H, int W, int64_t stride_bytes, const double Q[16], void *const *d_disp_i16, void *const *d_disp_f32,
                     void *const *d_xyz_f32, const hipEvent_t *in_ready, const hipEvent_t *in_used, const hipEvent_t *out_done)
{
    int rc;
    // cost stage of every pair on the stream of its own engine, from where `e`'s stream stands now (the caller's
    // inputs may have been produced on it).  Side by side rather than one after the other: the per-pixel cost
    // kernel is bound by the vector units, the box filter by HBM -- pairs in different kernels overlap (12 pairs
    // 4K D=256: 7.2 against 7.4 ms per pair with the cost stages in one stream).
    HIP_TRY(hipEventRecord(e->ev_group, e->stream));
    for (int k = 1; k < n; k++) HIP_TRY(hipStreamWaitEvent(eng[k]->stream, e->ev_group, 0));
    for (int k = 0; k < n; k++) {
        if (in_ready && in_ready[k]) HIP_TRY(hipStreamWaitEvent(eng[k]->stream, in_ready[k], 0));
        if ((rc = run_compute(eng[k], (const uint8_t *)d_left[k], (const uint8_t *)d_right[k], H, W, stride_bytes,
                              (int16_t *)d_disp_i16[k], PH_PRE)))
            return rc;
        if (!eng[k]->plan_chain) return set_err(SGM_ERR_HIP, "internal: a pair of a chained group did not plan a chained sweep");
        if (in_used && in_used[k]) HIP_TRY(hipEventRecord(in_used[k], eng[k]->stream));   // (behind the whole cost stage: the images are read by its first kernel only)
    }
    // ---- the sweeps of all n pairs: one launch per pass on e's stream, behind every pair's cost stage
    const Geom &g = e->g;
    const int R = pl.R, nbands = pl.nbands, npass = pl.npass;
    const size_t ctl_bytes = ((size_t)(1 + (size_t)n * nbands) * 4 + 15) & ~(size_t)15;
    if ((rc = e->chain_ctl.ensure(ctl_bytes))) return rc;
    for (int k = 1; k < n; k++) {
        HIP_TRY(hipEventRecord(eng[k]->ev_done, eng[k]->stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, eng[k]->ev_done, 0));
    }
    stage_break(e);
    ChainFrames fr;
    fr.nf = n;
    for (int k = 0; k < n; k++) {
        fr.C[k] = (const int16_t *)eng[k]->cost.p;
        fr.S[k] = (int16_t *)eng[k]->aggr.p;
        fr.bnd[k] = (int16_t *)eng[k]->bndL.p;
        fr.hr[k] = eng[k]->g.hr;
    }
    for (int pass = 0; pass < npass; pass++) {
        const int ydir = pass == 0 ? 1 : -1;
        SweepArgs a{ydir, ydir, R, nullptr, nullptr, nullptr, nullptr, 0, e->debug, (uint32_t *)e->chain_ctl.p, nullptr,
                    (uint32_t *)e->chain_err.p, nbands};
        HIP_TRY(hipMemsetAsync(e->chain_ctl.p, 0, ctl_bytes, e->stream));
        stage_break(e);
        if ((rc = stage_begin(e, pass == 0 ? "chain_dn" : "chain_up"))) return rc;
        if ((rc = launch_chain(g, a, fr, pass == 0 ? SWEEP_FIRST : SWEEP_ACCUM, chain_window(g, R, nbands, n, e->chain_wgs), e->stream)))
            return rc;
        KCHECK();
        if ((rc = stage_end(e, 1))) return rc;
    }
    HIP_TRY(hipEventRecord(e->ev_group, e->stream));
    // ---- the rest of every pair on its own stream (memory-bound kernels of different pairs side by side)
    for (int k = 0; k < n; k++) {
        if (k > 0) HIP_TRY(hipStreamWaitEvent(eng[k]->stream, e->ev_group, 0));
        // Host entry: the epilogues three at a time, not all at once -- they are bound by HBM either way, and pair k's map
        // can travel to the host while the epilogues of the pairs behind it still run (all n at once end together: the
        // whole group's download would follow the last kernel).
        if (out_done && k >= 3 && out_done[k - 3]) HIP_TRY(hipStreamWaitEvent(eng[k]->stream, out_done[k - 3], 0));
        if ((rc = run_compute(eng[k], (const uint8_t *)d_left[k], (const uint8_t *)d_right[k], H, W, stride_bytes,
                              (int16_t *)d_disp_i16[k], PH_POST)))
            return rc;
        if ((rc = run_float_xyz(eng[k], (const int16_t *)d_disp_i16[k], H, W, Q, d_disp_f32 ? d_disp_f32[k] : nullptr,
                                d_xyz_f32 ? d_xyz_f32[k] : nullptr)))
            return rc;
        if (out_done && out_done[k]) HIP_TRY(hipEventRecord(out_done[k], eng[k]->stream));
    }
    // e's stream ends behind everything the group did
    for (int k = 1; k < n; k++) {
        HIP_TRY(hipEventRecord(eng[k]->ev_done, eng[k]->stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, eng[k]->ev_done, 0));
    }
    stage_break(e);
    return SGM_OK;
}

// can this call run chained, and with which plan ?  (decided from the geometry alone: nothing is created or enqueued)
static int batch_plan(sgm_engine *e, int N, int H, int W, Plan *pl, bool *joint)
{
    if (H <= 0 || W < 2) return set_err(SGM_ERR_INVALID_ARG, "bad shape H=%d W=%d", H, W);
    if (W > 32767 || H > 32767) return set_err(SGM_ERR_UNSUPPORTED, "image larger than 32767 in a dimension");
    Geom g;
    int rc = normalise(&e->params, H, W, &g);
    if (rc) return rc;
    *pl = make_plan(e, g, H);
    *joint = e->schedule == 2 && N > 1 && pl->chain && e->group_max != 1;
    return SGM_OK;
}

// N pairs resident in device memory.  Chained groups when the configuration allows (groups as large as device memory
// holds, up to CHAIN_MAX_FRAMES or SGM_OPT_GROUP_MAX; a batch larger than a group is cut into groups of equal size);
// otherwise pair after pair on `e` (the schedule `e` is set to).  Results equal N calls of sgm_pipeline_device.
// Asynchronous like sgm_pipeline_device: returns when everything is enqueued; sgm_synchronize(e) waits for all of it.
// On an error every stream of the group is drained before the call returns.
int sgm_pipeline_batch_device(sgm_engine *e, int N, const void *const *d_left, const void *const *d_right, int H, int W,
                              int64_t stride_bytes, const double Q[16], void *const *d_disp_i16, void *const *d_disp_f32,
                              void *const *d_xyz_f32)
{
    if (!e || N <= 0 || !d_left || !d_right || !d_disp_i16) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    for (int i = 0; i < N; i++)
        if (!d_left[i] || !d_right[i] || !d_disp_i16[i]) return set_err(SGM_ERR_INVALID_ARG, "null pointer for pair %d", i);
    if (stride_bytes < W) return set_err(SGM_ERR_INVALID_ARG, "bad shape H=%d W=%d stride=%lld", H, W, (long long)stride_bytes);
    if (d_xyz_f32 && !Q) return set_err(SGM_ERR_INVALID_ARG, "Q is null");
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    Plan pl;
    bool joint = false;
    if ((rc = batch_plan(e, N, H, W, &pl, &joint))) return rc;
    BatchGuard guard{e};
    e->last_group = 0;
    int cap = 1;
    if (joint) {
        if ((rc = prepare_group(e, N, H, W, pl, &cap))) return rc;
        if (!e->chain_err.p) {
            if ((rc = e->chain_err.ensure(16))) return rc;
            HIP_TRY(hipMemsetAsync(e->chain_err.p, 0, 16, e->stream));
        }
    }
    if (!joint || cap < 2) {
        for (int i = 0; i < N; i++) {
            e->hr_accumulate = i > 0;     // (the headroom record of the call covers every pair)
            if ((rc = sgm_pipeline_device(e, d_left[i], d_right[i], H, W, stride_bytes, Q, d_disp_i16[i],
                                          d_disp_f32 ? d_disp_f32[i] : nullptr, d_xyz_f32 ? d_xyz_f32[i] : nullptr)))
                return rc;
        }
        guard.ok = true;
        return SGM_OK;
    }
    const int ngroups = (N + cap - 1) / cap, per = (N + ngroups - 1) / ngroups;   // groups of equal size
    e->last_group = std::min(per, N) - 1;
    std::vector<sgm_engine *> eng(per);
    for (int k = 0; k < per; k++) eng[k] = k == 0 ? e : e->group[k - 1];
    for (int i0 = 0; i0 < N; i0 += per) {
        const int n = std::min(N - i0, per);
        for (int k = 0; k < n; k++) eng[k]->hr_accumulate = i0 > 0;
        if (n == 1) {
            if ((rc = sgm_pipeline_device(e, d_left[i0], d_right[i0], H, W, stride_bytes, Q, d_disp_i16[i0],
                                          d_disp_f32 ? d_disp_f32[i0] : nullptr, d_xyz_f32 ? d_xyz_f32[i0] : nullptr)))
                return rc;
            continue;
        }
        if ((rc = run_group(e, eng.data(), n, pl, d_left + i0, d_right + i0, H, W, stride_bytes, Q, d_disp_i16 + i0,
                            d_disp_f32 ? d_disp_f32 + i0 : nullptr, d_xyz_f32 ? d_xyz_f32 + i0 : nullptr, nullptr, nullptr, nullptr)))
            return rc;
    }
    guard.ok = true;
    return SGM_OK;
}

int sgm_compute(sgm_engine *e, const uint8_t *left, const uint8_t *right, int H, int W, int64_t stride_bytes,
                int16_t *disp_out)
{
    if (!e || !left || !right || !disp_out) return set_err(SGM_ERR_INVALID_ARG, "null pointer");
    if (H <= 0 || W < 2 || stride_bytes < W) return set_err(SGM_ERR_INVALID_ARG, "bad shape H=%d W=%d stride=%lld", H, W, (long long)stride_bytes);
    HIP_TRY(hipSetDevice(e->device));
    const size_t npx = (size_t)H * W;
    int rc;
    if ((rc = e->in_left.ensure(npx)) || (rc = e->in_right.ensure(npx)) || (rc = e->disp_out.ensure(npx * 2))) return rc;
    HIP_TRY(hipMemcpy2DAsync(e->in_left.p, W, left, (size_t)stride_bytes, W, H, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpy2DAsync(e->in_right.p, W, right, (size_t)stride_bytes, W, H, hipMemcpyHostToDevice, e->stream));
    if ((rc = run_compute(e, (const uint8_t *)e->in_left.p, (const uint8_t *)e->in_right.p, H, W, W, (int16_t *)e->disp_out.p))) return rc;
    HIP_TRY(hipMemcpyAsync(disp_out, e->disp_out.p, npx * 2, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return check_chain(e);
}

// memcpy between pageable and page-locked host memory with a few threads (large blocks only: one thread moves about
// 10 GB/s, a 4K frame is 8 - 17 MB)
static void host_copy(void *dst, const void *src, size_t bytes)
{
    const int nt = bytes >= ((size_t)4 << 20) ? 4 : 1;
    if (nt == 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t part = (bytes / nt + 4095) & ~(size_t)4095;
    std::thread th[3];
    for (int t = 1; t < nt; t++) {
        const size_t o = std::min(bytes, part * t), n = std::min(bytes - o, part);
        th[t - 1] = std::thread([=] { std::memcpy((char *)dst + o, (const char *)src + o, n); });
    }
    std::memcpy(dst, src, std::min(bytes, part));
    for (int t = 1; t < nt; t++) th[t - 1].join();
}

// N independent pairs from / to host memory.  Up to three pairs are in flight: pair i runs on engine
// i % 3 (the engine itself and two peers with their own streams and device buffers, created on first
// use).  Images and disparity maps are staged through page-locked buffers of the engine they run on, so
// every transfer is asynchronous: while the host copies the results of pair i - 3 out of, and pair i
// into, the staging buffers of one engine, the other two keep the GPU busy with two frames (a second
// frame fills the issue slots and the HBM time one frame leaves idle).  The XYZ image
// (99.5 MB per 4K pair) goes straight to the caller's buffer: staging it would cost the host more than
// the pageable copy does.
int sgm_compute_batch(sgm_engine *e, int N, const uint8_t *lefts, const uint8_t *rights, int H, int W,
                      int16_t *disps_out, float *xyz_out, const double *Q16)
{
    if (!e || !lefts || !rights || !disps_out || N <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (xyz_out && !Q16) return set_err(SGM_ERR_INVALID_ARG, "xyz_out requested without Q");
    if (H <= 0 || W < 2) return set_err(SGM_ERR_INVALID_ARG, "bad shape");
    HIP_TRY(hipSetDevice(e->device));
    const size_t npx = (size_t)H * W;
    int rc;
    Plan pl;
    bool joint = false;
    if ((rc = batch_plan(e, N, H, W, &pl, &joint))) return rc;
    if (joint) {
        // Throughput mode: chained groups as large as device memory holds, TWO groups in flight -- while the kernels of
        // group g run, the images of group g + 1 are uploaded and the maps of group g - 1 downloaded on two copy streams
        // of their own.  Every pair has its device images twice (slot g & 1) on the engine that runs it, with page-locked
        // staging beside them (copies from / to pageable memory would block the host until the GPU gets round to them --
        // and a chained sweep launch holds every CU for its whole length), and four events:
        //   in:   images uploaded   -- the pair's cost stage waits for this one only: the first kernels start when the first
        //                              pair has arrived, not the whole group
        //   used: images consumed   -- the upload of group g + 2 into the same slot waits for it
        //   out:  last kernel done  -- the pair's download waits for this one only (the epilogues of a group run three at a
        //                              time, so maps leave while the pairs behind them are still being finished)
        //   dl:   map in the staging buffer -- the host copies it to the caller's array; passed before the slot is reused
        // The host copies caller -> staging -> caller with a few threads (one thread moves about 10 GB/s; 17 4K maps are 282 MB).
        // The XYZ images (99.5 MB each) go straight to the caller's memory: staging them would pin gigabytes.
        BatchGuard guard{e};
        int cap = 1;
        const size_t slot_bytes = 2 * (2 * npx + npx * 2 + (xyz_out ? npx * 16 : 0));
        if ((rc = prepare_group(e, N, H, W, pl, &cap, slot_bytes))) return rc;
        if (cap >= 2) {
            if (!e->chain_err.p) {
                if ((rc = e->chain_err.ensure(16))) return rc;
                HIP_TRY(hipMemsetAsync(e->chain_err.p, 0, 16, e->stream));
            }
            if (!e->copy_in) HIP_TRY(hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking));
            if (!e->copy_out) HIP_TRY(hipStreamCreateWithFlags(&e->copy_out, hipStreamNonBlocking));
            const int ngroups = (N + cap - 1) / cap, per = (N + ngroups - 1) / ngroups;
            e->last_group = per - 1;
            std::vector<sgm_engine *> eng(per);
            for (int k = 0; k < per; k++) {
                sgm_engine *q = eng[k] = k == 0 ? e : e->group[k - 1];
                for (int sl = 0; sl < (ngroups > 1 ? 2 : 1); sl++) {
                    if ((rc = q->io[sl][0].ensure(npx)) || (rc = q->io[sl][1].ensure(npx)) || (rc = q->io[sl][2].ensure(npx * 2))) return rc;
                    if ((rc = q->pin_io[sl][0].ensure(npx)) || (rc = q->pin_io[sl][1].ensure(npx)) || (rc = q->pin_io[sl][2].ensure(npx * 2))) return rc;
                    if (xyz_out && ((rc = q->io[sl][3].ensure(npx * 4)) || (rc = q->io[sl][4].ensure(npx * 12)))) return rc;
                    for (hipEvent_t *ev : {&q->ev_io_in[sl], &q->ev_io_used[sl], &q->ev_io_out[sl], &q->ev_io_dl[sl]})
                        if (!*ev) HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
                }
            }
            std::vector<const void *> dl(per), dr(per);
            std::vector<void *> dd(per), df(per), dx(per);
            std::vector<hipEvent_t> evi(per), evu(per), evo(per);
            auto upload = [&](int gi) -> int {     // images of group gi -> slot gi & 1, pair by pair
                const int sl = gi & 1, i0 = gi * per, n = std::min(N - i0, per);
                for (int k = 0; k < n; k++) {
                    sgm_engine *q = eng[k];
                    const size_t i = (size_t)(i0 + k);
                    if (gi >= 2) {
                        HIP_TRY(hipEventSynchronize(q->ev_io_in[sl]));                 // the staging buffers: their last upload has left them (long ago)
                        HIP_TRY(hipStreamWaitEvent(e->copy_in, q->ev_io_used[sl], 0));  // the device images: group gi - 2 has read them
                    }
                    host_copy(q->pin_io[sl][0].p, lefts + i * npx, npx);
                    host_copy(q->pin_io[sl][1].p, rights + i * npx, npx);
                    HIP_TRY(hipMemcpyAsync(q->io[sl][0].p, q->pin_io[sl][0].p, npx, hipMemcpyHostToDevice, e->copy_in));
                    HIP_TRY(hipMemcpyAsync(q->io[sl][1].p, q->pin_io[sl][1].p, npx, hipMemcpyHostToDevice, e->copy_in));
                    HIP_TRY(hipEventRecord(q->ev_io_in[sl], e->copy_in));
                }
                return SGM_OK;
            };
            auto compute = [&](int gi) -> int {    // kernels of group gi, and -- behind each pair's last one -- its download
                const int sl = gi & 1, i0 = gi * per, n = std::min(N - i0, per);
                for (int k = 0; k < n; k++) {
                    sgm_engine *q = eng[k];
                    q->hr_accumulate = gi > 0;
                    dl[k] = q->io[sl][0].p;
                    dr[k] = q->io[sl][1].p;
                    dd[k] = q->io[sl][2].p;
                    df[k] = xyz_out ? q->io[sl][3].p : nullptr;
                    dx[k] = xyz_out ? q->io[sl][4].p : nullptr;
                    evi[k] = q->ev_io_in[sl];
                    evu[k] = q->ev_io_used[sl];
                    evo[k] = q->ev_io_out[sl];
                }
                if (n == 1) {   // (a last group of one pair: the plain entry on e, behind its upload)
                    HIP_TRY(hipStreamWaitEvent(e->stream, evi[0], 0));
                    int r2 = sgm_pipeline_device(e, dl[0], dr[0], H, W, W, Q16, dd[0], df[0], dx[0]);
                    if (r2) return r2;
                    HIP_TRY(hipEventRecord(evu[0], e->stream));
                    HIP_TRY(hipEventRecord(evo[0], e->stream));
                } else {
                    int r2 = run_group(e, eng.data(), n, pl, dl.data(), dr.data(), H, W, W, Q16, dd.data(), xyz_out ? df.data() : nullptr,
                                       xyz_out ? dx.data() : nullptr, evi.data(), evu.data(), evo.data());
                    if (r2) return r2;
                }
                for (int k = 0; k < n; k++) {
                    sgm_engine *q = eng[k];
                    HIP_TRY(hipStreamWaitEvent(e->copy_out, q->ev_io_out[sl], 0));
                    HIP_TRY(hipMemcpyAsync(q->pin_io[sl][2].p, q->io[sl][2].p, npx * 2, hipMemcpyDeviceToHost, e->copy_out));
                    HIP_TRY(hipEventRecord(q->ev_io_dl[sl], e->copy_out));
                }
                return SGM_OK;
            };
            auto finish = [&](int gi) -> int {     // maps (and XYZ) of group gi into the caller's arrays
                const int sl = gi & 1, i0 = gi * per, n = std::min(N - i0, per);
                for (int k = 0; k < n; k++) {
                    sgm_engine *q = eng[k];
                    const size_t i = (size_t)(i0 + k);
                    HIP_TRY(hipEventSynchronize(q->ev_io_dl[sl]));
                    host_copy(disps_out + i * npx, q->pin_io[sl][2].p, npx * 2);
                    if (xyz_out) HIP_TRY(hipMemcpy(xyz_out + i * npx * 3, q->io[sl][4].p, npx * 12, hipMemcpyDeviceToHost));
                }
                return SGM_OK;
            };
            if ((rc = upload(0))) return rc;
            for (int gi = 0; gi < ngroups; gi++) {
                if ((rc = compute(gi))) return rc;
                if (gi + 1 < ngroups && (rc = upload(gi + 1))) return rc;     // beside the kernels of group gi
                if (gi > 0 && (rc = finish(gi - 1))) return rc;                // (its slot's maps are rewritten by group gi + 1, enqueued after this)
            }
            if ((rc = finish(ngroups - 1))) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            guard.ok = true;
            return check_chain(e);
        }
        // (not even two pairs fit beside each other: pair after pair below)
    }
    const int neng = std::min(N, 3);
    if (neng > 1 && !e->peer && (rc = sgm_create(&e->params, e->device, nullptr, &e->peer))) return rc;
    if (neng > 2 && !e->peer2 && (rc = sgm_create(&e->params, e->device, nullptr, &e->peer2))) return rc;
    sgm_engine *eng[3] = {e, e->peer, e->peer2};
    const int saved_keep = e->keep_aggr, saved_profile = e->profile;
    // every exit drains the streams of all engines (copies into / out of the page-locked staging buffers and kernels may
    // still be in flight when an error is returned) and gives the caller's engine its own options back
    struct Drain {
        sgm_engine **eng;
        int n, keep, prof;
        ~Drain()
        {
            for (int k = 0; k < n; k++)
                if (eng[k]) (void)hipStreamSynchronize(eng[k]->stream);
            eng[0]->keep_aggr = keep;
            eng[0]->profile = prof;
        }
    } drain{eng, neng, saved_keep, saved_profile};
    for (int k = 0; k < neng; k++) {
        sgm_engine *q = eng[k];
        q->keep_aggr = 0;
        q->profile = 0;
        q->schedule = e->schedule;
        q->sweep_rows = e->sweep_rows;
        q->debug = e->debug;
        q->prepass_rows = e->prepass_rows;
        if ((rc = q->in_left.ensure(npx)) || (rc = q->in_right.ensure(npx)) || (rc = q->disp_out.ensure(npx * 2))) return rc;
        if ((rc = q->pin_left.ensure(npx)) || (rc = q->pin_right.ensure(npx)) || (rc = q->pin_disp.ensure(npx * 2))) return rc;
        if (xyz_out && ((rc = q->f32.ensure(npx * 4)) || (rc = q->xyz.ensure(npx * 12)))) return rc;
        if (!q->ev_done) HIP_TRY(hipEventCreateWithFlags(&q->ev_done, hipEventDisableTiming));
    }
    // results of pair i leave its engine right before the engine is reused for pair i + neng
    auto finish = [&](int i) -> int {
        sgm_engine *q = eng[i % neng];
        HIP_TRY(hipEventSynchronize(q->ev_done));
        std::memcpy(disps_out + (size_t)i * npx, q->pin_disp.p, npx * 2);
        if (xyz_out) HIP_TRY(hipMemcpy(xyz_out + (size_t)i * npx * 3, q->xyz.p, npx * 12, hipMemcpyDeviceToHost));
        return SGM_OK;
    };
    for (int i = 0; i < N; i++) {
        sgm_engine *q = eng[i % neng];
        if (i >= neng && (rc = finish(i - neng))) return rc;
        std::memcpy(q->pin_left.p, lefts + (size_t)i * npx, npx);
        std::memcpy(q->pin_right.p, rights + (size_t)i * npx, npx);
        HIP_TRY(hipMemcpyAsync(q->in_left.p, q->pin_left.p, npx, hipMemcpyHostToDevice, q->stream));
        HIP_TRY(hipMemcpyAsync(q->in_right.p, q->pin_right.p, npx, hipMemcpyHostToDevice, q->stream));
        rc = sgm_pipeline_device(q, q->in_left.p, q->in_right.p, H, W, W, Q16, q->disp_out.p, xyz_out ? q->f32.p : nullptr,
                                 xyz_out ? q->xyz.p : nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(q->pin_disp.p, q->disp_out.p, npx * 2, hipMemcpyDeviceToHost, q->stream));
        HIP_TRY(hipEventRecord(q->ev_done, q->stream));
    }
    for (int i = std::max(0, N - neng); i < N; i++)
        if ((rc = finish(i))) return rc;
    return SGM_OK;
}

int sgm_disp_to_float(sgm_engine *e, const int16_t *disp, int64_t n, float *out)
{
    if (!e || !disp || !out || n <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if ((rc = e->disp_out.ensure((size_t)n * 2)) || (rc = e->f32.ensure((size_t)n * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(e->disp_out.p, disp, (size_t)n * 2, hipMemcpyHostToDevice, e->stream));
    if ((rc = run_to_float(e, (const int16_t *)e->disp_out.p, n, (float *)e->f32.p))) return rc;
    HIP_TRY(hipMemcpyAsync(out, e->f32.p, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_reproject(sgm_engine *e, const float *disp, int H, int W, const double Q[16], int handle_missing, float *xyz_out)
{
    if (!e || !disp || !xyz_out || !Q || H <= 0 || W <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t npx = (size_t)H * W;
    int rc;
    if ((rc = e->f32.ensure(npx * 4)) || (rc = e->xyz.ensure(npx * 12))) return rc;
    HIP_TRY(hipMemcpyAsync(e->f32.p, disp, npx * 4, hipMemcpyHostToDevice, e->stream));
    if ((rc = run_reproject(e, (const float *)e->f32.p, H, W, Q, handle_missing, (float *)e->xyz.p))) return rc;
    HIP_TRY(hipMemcpyAsync(xyz_out, e->xyz.p, npx * 12, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_valid_mask(sgm_engine *e, const float *xyz, const float *disp, int64_t n, uint8_t *mask)
{
    if (!e || !xyz || !disp || !mask || n <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if ((rc = e->f32.ensure((size_t)n * 4)) || (rc = e->xyz.ensure((size_t)n * 12)) || (rc = e->mask.ensure((size_t)n))) return rc;
    HIP_TRY(hipMemcpyAsync(e->f32.p, disp, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->xyz.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, e->stream));
    if ((rc = sgm_valid_mask_device(e, e->xyz.p, e->f32.p, n, e->mask.p))) return rc;
    HIP_TRY(hipMemcpyAsync(mask, e->mask.p, (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

// the three launches of an ordered compaction on the engine's stream; the total stays in e->ccount[m]
static int enqueue_compaction(sgm_engine *e, const void *d_xyz, const void *d_disp_f32, const void *d_colors_rgb, int64_t n,
                              void *d_out_points, void *d_out_colors, int *m_out)
{
    if (!e || !d_xyz || !d_disp_f32 || !d_out_points || n <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (d_out_colors && !d_colors_rgb) return set_err(SGM_ERR_INVALID_ARG, "out_colors requested without colors");
    if (n >= (1ll << 32)) return set_err(SGM_ERR_UNSUPPORTED, "more than 2^32 points");
    HIP_TRY(hipSetDevice(e->device));
    const int m = (int)((n + 255) / 256);
    int rc;
    if ((rc = e->ccount.ensure((size_t)(m + 1) * 4))) return rc;
    uint32_t *cnt = (uint32_t *)e->ccount.p;
    hipStream_t st = e->stream;
    hipLaunchKernelGGL(k_compact_count, dim3(m), dim3(256), 0, st, (const float *)d_xyz, (const float *)d_disp_f32, n, cnt);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, cnt, m);
    hipLaunchKernelGGL(k_compact_scatter, dim3(m), dim3(256), 0, st, (const float *)d_xyz, (const float *)d_disp_f32,
                       (const uint8_t *)(d_out_colors ? d_colors_rgb : nullptr), n, (const uint32_t *)cnt,
                       (float *)d_out_points, (uint8_t *)d_out_colors);
    KCHECK();
    *m_out = m;
    return SGM_OK;
}

int sgm_compact_points_device(sgm_engine *e, const void *d_xyz, const void *d_disp_f32, const void *d_colors_rgb,
                              int64_t n, void *d_out_points, void *d_out_colors, int64_t *n_valid)
{
    if (!n_valid) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    int m = 0;
    int rc = enqueue_compaction(e, d_xyz, d_disp_f32, d_colors_rgb, n, d_out_points, d_out_colors, &m);
    if (rc) return rc;
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, (uint32_t *)e->ccount.p + m, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *n_valid = (int64_t)total;
    return SGM_OK;
}

// the same without a host round trip: the count goes to device memory (stream order), nothing is synchronised
int sgm_compact_points_device_async(sgm_engine *e, const void *d_xyz, const void *d_disp_f32, const void *d_colors_rgb,
                                    int64_t n, void *d_out_points, void *d_out_colors, void *d_n_valid_i64)
{
    if (!d_n_valid_i64) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    int m = 0;
    int rc = enqueue_compaction(e, d_xyz, d_disp_f32, d_colors_rgb, n, d_out_points, d_out_colors, &m);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact_total, dim3(1), dim3(1), 0, e->stream, (const uint32_t *)e->ccount.p + m, (int64_t *)d_n_valid_i64);
    KCHECK();
    return SGM_OK;
}

int sgm_compact_points(sgm_engine *e, const float *xyz, const float *disp, const uint8_t *colors_rgb, int64_t n,
                       float *out_points, uint8_t *out_colors, int64_t *n_valid)
{
    if (!e || !xyz || !disp || !out_points || !n_valid || n <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (out_colors && !colors_rgb) return set_err(SGM_ERR_INVALID_ARG, "out_colors requested without colors");
    HIP_TRY(hipSetDevice(e->device));
    int rc;
    if ((rc = e->f32.ensure((size_t)n * 4)) || (rc = e->xyz.ensure((size_t)n * 12)) || (rc = e->cpts.ensure((size_t)n * 12))) return rc;
    if (out_colors && ((rc = e->crgb_in.ensure((size_t)n * 3)) || (rc = e->crgb.ensure((size_t)n * 3)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->f32.p, disp, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->xyz.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, e->stream));
    if (out_colors) HIP_TRY(hipMemcpyAsync(e->crgb_in.p, colors_rgb, (size_t)n * 3, hipMemcpyHostToDevice, e->stream));
    if ((rc = sgm_compact_points_device(e, e->xyz.p, e->f32.p, out_colors ? e->crgb_in.p : nullptr, n, e->cpts.p,
                                        out_colors ? e->crgb.p : nullptr, n_valid)))
        return rc;
    if (*n_valid > 0) {
        HIP_TRY(hipMemcpyAsync(out_points, e->cpts.p, (size_t)*n_valid * 12, hipMemcpyDeviceToHost, e->stream));
        if (out_colors) HIP_TRY(hipMemcpyAsync(out_colors, e->crgb.p, (size_t)*n_valid * 3, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_median3x3(sgm_engine *e, const int16_t *src, int H, int W, int16_t *dst)
{
    if (!e || !src || !dst || H <= 0 || W <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t npx = (size_t)H * W;
    int rc;
    if ((rc = e->disp_raw.ensure(npx * 2)) || (rc = e->disp_med.ensure(npx * 2))) return rc;
    HIP_TRY(hipMemcpyAsync(e->disp_raw.p, src, npx * 2, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_median3, dim3((W + 255) / 256, H), dim3(256), 0, e->stream, (const int16_t *)e->disp_raw.p,
                       (int16_t *)e->disp_med.p, (int16_t *)nullptr, H, W);
    KCHECK();
    HIP_TRY(hipMemcpyAsync(dst, e->disp_med.p, npx * 2, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_filter_speckles(sgm_engine *e, int16_t *img, int H, int W, int newVal, int maxSpeckleSize, int maxDiff)
{
    if (!e || !img || H <= 0 || W <= 0) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t npx = (size_t)H * W;
    int rc;
    if ((rc = e->disp_out.ensure(npx * 2))) return rc;
    HIP_TRY(hipMemcpyAsync(e->disp_out.p, img, npx * 2, hipMemcpyHostToDevice, e->stream));
    if ((rc = run_speckles(e, (int16_t *)e->disp_out.p, H, W, newVal, maxSpeckleSize, maxDiff))) return rc;
    HIP_TRY(hipMemcpyAsync(img, e->disp_out.p, npx * 2, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SGM_OK;
}

int sgm_get_tap(sgm_engine *e, int tap, void *host_dst, int64_t bytes)
{
    if (!e || !host_dst) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (e->H <= 0) return set_err(SGM_ERR_INVALID_ARG, "no compute has run yet");
    HIP_TRY(hipSetDevice(e->device));
    const int64_t npx = (int64_t)e->H * e->W;
    const int64_t vol = std::max<int64_t>(e->g.rowsz, 0) * e->H * 2;
    const void *src = nullptr;
    int64_t need = 0;
    switch (tap) {
    case SGM_TAP_COST: src = e->cost.p; need = vol; break;
    case SGM_TAP_AGGR:
        if (!e->keep_aggr) return set_err(SGM_ERR_INVALID_ARG, "SGM_TAP_AGGR needs SGM_OPT_KEEP_AGGR=1 before compute");
        src = e->aggr.p; need = vol; break;
    case SGM_TAP_DISP_RAW: src = e->disp_raw.p; need = npx * 2; break;
    case SGM_TAP_DISP_MEDIAN: src = e->disp_med.p; need = npx * 2; break;
    default: return set_err(SGM_ERR_INVALID_ARG, "unknown tap %d", tap);
    }
    if (bytes != need) return set_err(SGM_ERR_INVALID_ARG, "tap %d holds %lld bytes, caller passed %lld", tap, (long long)need, (long long)bytes);
    if (need == 0) return SGM_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_dst, src, (size_t)need, hipMemcpyDeviceToHost));
    return check_chain(e);
}

int sgm_get_headroom(sgm_engine *e, int *max_cost_plus_p2, int *max_delta, int *ok)
{
    if (!e) return set_err(SGM_ERR_INVALID_ARG, "engine is null");
    if (e->H <= 0 || !e->headroom.p) return set_err(SGM_ERR_INVALID_ARG, "no compute has run yet");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // the last call's record: after a batch call, the maximum over every pair of the batch (each pair of a chained group
    // keeps its record on the internal engine that ran it; e's stream ends behind all of them)
    uint32_t h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, e->headroom.p, 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < e->last_group && k < (int)e->group.size(); k++) {
        uint32_t q[2] = {0, 0};
        if (!e->group[k]->headroom.p) continue;
        HIP_TRY(hipMemcpy(q, e->group[k]->headroom.p, 8, hipMemcpyDeviceToHost));
        h[0] = std::max(h[0], q[0]);
        h[1] = std::max(h[1], q[1]);
    }
    // an int16 lane upstream holds C_true + P2 and min_d L_r + P2: both must fit
    const int64_t a = e->g.W1 > 0 ? (int64_t)h[0] + e->g.P2 : 0, b = e->g.W1 > 0 ? (int64_t)h[1] + e->g.P2 : 0;
    if (max_cost_plus_p2) *max_cost_plus_p2 = (int)std::min<int64_t>(a, INT32_MAX);
    if (max_delta) *max_delta = (int)std::min<int64_t>(b, INT32_MAX);
    if (ok) *ok = (a <= SGM_MAX_COST && b <= SGM_MAX_COST) ? 1 : 0;
    return SGM_OK;
}

int sgm_get_stage_times(sgm_engine *e, sgm_stage_times *out)
{
    if (!e || !out) return set_err(SGM_ERR_INVALID_ARG, "bad argument");
    if (!e->profile) return set_err(SGM_ERR_INVALID_ARG, "SGM_OPT_PROFILE is off");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // one entry per stage NAME: a stage may consist of several bracketed launches (row chunks of the
    // pipelined first pass, on streams of their own) -- their HIP-event times and launch counts add up
    out->n = 0;
    for (int i = 0; i < e->nstages; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->events[e->stage_ev[i * 2]], e->events[e->stage_ev[i * 2 + 1]]));
        int k = 0;
        while (k < out->n && strcmp(out->name[k], e->stage_names[i]) != 0) k++;
        if (k == out->n) {
            if (out->n >= SGM_MAX_STAGES - 1) continue;
            out->name[k] = e->stage_names[i];
            out->ms[k] = 0.f;
            out->launches[k] = 0;
            out->n++;
        }
        out->ms[k] += ms;
        out->launches[k] += e->stage_launches[i];
    }
    // stages overlap (auxiliary / chunk streams): also report first-begin -> last-end of the main stream
    if (e->nstages > 0) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->events[e->stage_ev[0]], e->events[e->stage_ev[(e->nstages - 1) * 2 + 1]]));
        out->name[out->n] = "_wall";
        out->ms[out->n] = ms;
        out->launches[out->n] = 0;
        out->n++;
    }
    return SGM_OK;
}

// SURVEY.md 8(d):  B_alg = 2HW (L,R in) + V (1 + 3 Np) + 2HW (disp out) + 8HW (median, speckle r+w)
//                          [+ 16 HW reproject], V = 2 H W1 D bytes, Np = 5 (mode 0) or 8 (mode 1).
int64_t sgm_algorithmic_bytes(const sgm_params *params, int H, int W, int with_reproject)
{
    Geom g;
    if (!params || normalise(params, H, W, &g)) return -1;
    const int64_t HW = (int64_t)H * W;
    const int64_t V = 2 * (int64_t)H * std::max(g.W1, 0) * g.D;
    const int Np = g.mode == 1 ? 8 : 5;
    int64_t b = 2 * HW + V * (1 + 3 * Np) + 2 * HW + 8 * HW;
    if (with_reproject) b += 16 * HW;
    return b;
}

}  // extern "C"

// kernels_post.h -- everything after the aggregated cost: right-view map + sub-pixel + LR check,
// 3x3 median, speckle filter (parallel connected components), float scaling/masking,
// reprojection to XYZ and the validity mask.
//
// Replaces, in order: the tail of upstream computeDisparitySGBM, medianBlur(disp, disp, 3) and
// filterSpeckles inside stereo.compute (/root/reference/main.ipynb:668),
// the numpy lines main.ipynb:668-670, cv2.reprojectImageTo3D (main.ipynb:697; SURVEY.md
// Appendix B) and the mask of main.ipynb:726-730.  CPU restatement: oracle/sgbm_oracle.c.
#pragma once
#include "kernels_cost.h"
#include <float.h>

namespace sgm {

// ------------------------------------------------------------------------------------------
// One workgroup per image row.  The right-view disparity map of upstream ("disp2", filled in
// descending x, a candidate replaces the holder only with a strictly smaller cost) is the
// argmin over candidates of (cost, then larger x): an LDS atomicMin on (cost << 16 | 0xffff-x).
__global__ __launch_bounds__(256) void k_select(Geom g, const uint2 *__restrict__ wta,
                                                int16_t *__restrict__ disp)
{
    extern __shared__ uint32_t keys[];  // W entries
    const int y = blockIdx.x, W = g.W, maxX1 = g.minX1 + g.W1;
    const int INV = g.invalid_scaled;
    for (int x = threadIdx.x; x < W; x += blockDim.x) keys[x] = 0xffffffffu;
    __syncthreads();
    for (int x = g.minX1 + threadIdx.x; x < maxX1; x += blockDim.x) {
        const uint32_t k = wta[(int64_t)y * W + x].x;
        if (k != 0xffffffffu) {
            const int best = (int)(k & 0xffffu);
            const int x2 = x - best - g.minD;
            atomicMin(&keys[x2], (k & 0xffff0000u) | (uint32_t)(0xffff - x));
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int d1 = INV;
        if (x >= g.minX1 && x < maxX1) {
            const uint2 kv = wta[(int64_t)y * W + x];
            if (kv.x != 0xffffffffu) {
                const int best = (int)(kv.x & 0xffffu), s0 = (int)(kv.x >> 16);
                int dsc = best * 16;
                if (best > 0 && best < g.D - 1) {
                    const int sm = (int)(kv.y & 0xffffu), sp = (int)(kv.y >> 16);
                    const int denom2 = max(sm + sp - 2 * s0, 1);
                    dsc += ((sm - sp) * 16 + denom2) / (denom2 * 2);  // C division, toward zero
                }
                d1 = dsc + g.minD * 16;
            }
        }
        if (d1 != INV) {
            const int lo = d1 >> 4, hi = (d1 + 15) >> 4;
            const int xa = x - lo, xb = x - hi;
            bool kill = false;
            if (xa >= 0 && xa < W && xb >= 0 && xb < W) {
                const uint32_t ka = keys[xa], kb = keys[xb];
                const int da = ka == 0xffffffffu ? INV : (0xffff - (int)(ka & 0xffffu)) - xa;
                const int db = kb == 0xffffffffu ? INV : (0xffff - (int)(kb & 0xffffu)) - xb;
                kill = da >= g.minD && abs(da - lo) > g.d12 && db >= g.minD && abs(db - hi) > g.d12;
            }
            if (kill) d1 = INV;
        }
        disp[(int64_t)y * W + x] = (int16_t)d1;
    }
}

// ------------------------------------------------------------------------------------------
// Winner-take-all over the finished S volume, one LANE per pixel ("transposed" WTA).
//
// Inside the path kernels a pixel's D disparities are spread over the 64 lanes, so the argmin and
// the uniqueness test are wave reductions: ~110 instructions per pixel of a latency-bound kernel.
// Read back from HBM the same S vector can be scanned by a single lane with no cross-lane work:
// a block stages 64 consecutive pixels (64 * D * 2 bytes, one coalesced sweep) into LDS with a
// padded row stride, then every lane walks its own pixel twice --
//   pass 1: key = min over d of (S[d] << 16 | d)              -> minS and the FIRST best d
//   pass 2: far = min of S[d] over |d - best| > 1 (packed)    -> uniqueness (A.6 step 2)
// ~20 wave instructions per pixel, no dependent chain between pixels; the kernel is bound by
// reading V once.  Same record format as wta_pixels (kernels_path.h); same arithmetic as the
// selection loop of oracle/sgbm_oracle.c.
constexpr int wta_t_stride(int D) { return D * 2 + 8; }  // bytes per staged pixel row (8-byte aligned, breaks the bank stride)

// S = sat(S + S2), element-wise: only when the aggregated volume is kept for inspection (SGM_OPT_KEEP_AGGR)
// after a frame whose fifth path went into a volume of its own.
__global__ __launch_bounds__(256) void k_add_sat(int16_t *__restrict__ S, const int16_t *__restrict__ S2, int64_t n8)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    uint4 a = reinterpret_cast<uint4 *>(S)[i];
    const uint4 b = reinterpret_cast<const uint4 *>(S2)[i];
    a.x = pk_adds_s(a.x, b.x);
    a.y = pk_adds_s(a.y, b.y);
    a.z = pk_adds_s(a.z, b.z);
    a.w = pk_adds_s(a.w, b.w);
    reinterpret_cast<uint4 *>(S)[i] = a;
}

// LG = log2(chunks per row) when D is a power of two (16..512): compile-time trip counts, the next
// block's loads prefetched into registers during the current block's scan.  LG = -1: any D
// (multiple of 16), plain staging.
// NV: the costs are the saturating sum of NV volumes.  2: S + S2 (MODE_SGBM with D <= 128: the fifth path runs
// beside the sweep into a volume of its own); 3: S + S2 + S3 (D <= 64: both in-row paths run beside the per-row
// pre-pass and the element-wise vertical kernel, each into a volume of its own); 5: the five directions of MODE_SGBM each
// in a volume of its own (D <= 64: k_lines3_g + the two in-row paths).  Every path cost is >= 0 and the
// sum saturates, so the order of the additions does not matter -- kernels_path.h.
template <bool POSW, int LG, int NV = 1>
__global__ __launch_bounds__(64) void k_wta_t(Geom g, const int16_t *__restrict__ S, uint2 *__restrict__ wta, int64_t npix,
                                              const int16_t *__restrict__ S2 = nullptr, const int16_t *__restrict__ S3 = nullptr,
                                              const int16_t *__restrict__ S4 = nullptr, const int16_t *__restrict__ S5 = nullptr)
{
    constexpr bool TWO = NV >= 2, THREE = NV >= 3, FIVE = NV >= 5;
    static_assert(NV == 1 || NV == 2 || NV == 3 || NV == 5, "volumes: 1, 2, 3 or 5");
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];
    const int lane = threadIdx.x, D = LG >= 0 ? (8 << LG) : g.D, W1 = g.W1;
    const int stride = wta_t_stride(D);
    const int cpr = D * 2 / 16;  // 16-byte chunks per pixel row; a lane moves cpr chunks per block
    const int64_t nblocks = (npix + 63) / 64;
    constexpr int PF = LG < 0 ? 8 : (LG >= 5 ? 32 : (1 << LG));  // chunks per lane held in registers
    uint4 v[PF], v2[TWO ? PF : 1], v3[THREE ? PF : 1], v4[FIVE ? PF : 1], v5[FIVE ? PF : 1];
    // chunk c = lane + 64 k of the block's contiguous 64 * D * 2 bytes: loads with a clamped index
    // (no branch between them), committed to the padded LDS rows afterwards
    auto issue = [&](int64_t blk, int k0) {
        const int64_t left = npix - blk * 64;  // (integer compare: min<int64_t>() goes through v_min_f64)
        const int total = (left < 64 ? (int)left : 64) * cpr;
        const uint4 *src = reinterpret_cast<const uint4 *>(S + blk * 64 * D);
#pragma unroll
        for (int u = 0; u < PF; u++) v[u] = src[min(lane + 64 * (k0 + u), total - 1)];
        if constexpr (TWO) {
            const uint4 *src2 = reinterpret_cast<const uint4 *>(S2 + blk * 64 * D);
#pragma unroll
            for (int u = 0; u < PF; u++) v2[u] = src2[min(lane + 64 * (k0 + u), total - 1)];
        }
        if constexpr (THREE) {
            const uint4 *src3 = reinterpret_cast<const uint4 *>(S3 + blk * 64 * D);
#pragma unroll
            for (int u = 0; u < PF; u++) v3[u] = src3[min(lane + 64 * (k0 + u), total - 1)];
        }
        if constexpr (FIVE) {
            const uint4 *src4 = reinterpret_cast<const uint4 *>(S4 + blk * 64 * D);
            const uint4 *src5 = reinterpret_cast<const uint4 *>(S5 + blk * 64 * D);
#pragma unroll
            for (int u = 0; u < PF; u++) v4[u] = src4[min(lane + 64 * (k0 + u), total - 1)];
#pragma unroll
            for (int u = 0; u < PF; u++) v5[u] = src5[min(lane + 64 * (k0 + u), total - 1)];
        }
    };
    auto summed = [&](int u) {  // chunk u of the cost rows: S, or sat(S + S2)
        uint4 r = v[u];
        if constexpr (TWO) {
            r.x = pk_adds_s(r.x, v2[u].x);
            r.y = pk_adds_s(r.y, v2[u].y);
            r.z = pk_adds_s(r.z, v2[u].z);
            r.w = pk_adds_s(r.w, v2[u].w);
        }
        if constexpr (THREE) {
            r.x = pk_adds_s(r.x, v3[u].x);
            r.y = pk_adds_s(r.y, v3[u].y);
            r.z = pk_adds_s(r.z, v3[u].z);
            r.w = pk_adds_s(r.w, v3[u].w);
        }
        if constexpr (FIVE) {
            r.x = pk_adds_s(pk_adds_s(r.x, v4[u].x), v5[u].x);
            r.y = pk_adds_s(pk_adds_s(r.y, v4[u].y), v5[u].y);
            r.z = pk_adds_s(pk_adds_s(r.z, v4[u].z), v5[u].z);
            r.w = pk_adds_s(pk_adds_s(r.w, v4[u].w), v5[u].w);
        }
        return r;
    };
    auto commit = [&](int64_t blk, int k0) {
        const int64_t left = npix - blk * 64;
        const int total = (left < 64 ? (int)left : 64) * cpr;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int c = lane + 64 * (k0 + u);
            if (c < total) {
                const int px = LG >= 0 ? c >> LG : c / cpr, w = c - px * cpr;
                uint2 *dst = reinterpret_cast<uint2 *>(rows + px * stride + w * 16);
                const uint4 q = summed(u);
                dst[0] = make_uint2(q.x, q.y);
                dst[1] = make_uint2(q.z, q.w);
            }
        }
    };
    int64_t blk = blockIdx.x;
    if (LG >= 0 && blk < nblocks) issue(blk, 0);
    for (; blk < nblocks; blk += gridDim.x) {
    const int64_t p0 = blk * 64;
    const int np = npix - p0 < 64 ? (int)(npix - p0) : 64;
    __syncthreads();  // (one wave per block: orders the LDS traffic of consecutive blocks)
    if (LG >= 0) {
        commit(blk, 0);
        if (LG == 6) {  // D = 512: the second half of the rows, not prefetched
            issue(blk, PF);
            commit(blk, PF);
        }
    } else {
        for (int k0 = 0; k0 < cpr; k0 += PF) {  // cpr need not be a multiple of PF: clamped loads, guarded commits
            issue(blk, k0);
            const int total = np * cpr;
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int c = lane + 64 * (k0 + u);
                if (k0 + u < cpr && c < total) {
                    const int px = c / cpr, w = c - px * cpr;
                    uint2 *dst = reinterpret_cast<uint2 *>(rows + px * stride + w * 16);
                    const uint4 q = summed(u);
                    dst[0] = make_uint2(q.x, q.y);
                    dst[1] = make_uint2(q.z, q.w);
                }
            }
        }
    }
    __syncthreads();
    if (LG >= 0 && blk + gridDim.x < nblocks) issue(blk + gridDim.x, 0);  // next block's loads fly during this scan
    if (lane < np) {
    const uint8_t *row = rows + lane * stride;
    // pass 1
    uint32_t key = 0xffffffffu;
#pragma unroll 64  // fully unrolled for D <= 256: constant offsets and disparity indices, many LDS reads in flight
    for (int d0 = 0; d0 < D; d0 += 4) {
        const uint2 v = *reinterpret_cast<const uint2 *>(row + d0 * 2);
        const uint32_t k0 = (v.x << 16) | (uint32_t)d0, k1 = (v.x & 0xffff0000u) | (uint32_t)(d0 + 1);
        const uint32_t k2 = (v.y << 16) | (uint32_t)(d0 + 2), k3 = (v.y & 0xffff0000u) | (uint32_t)(d0 + 3);
        key = min(min(key, min(k0, k1)), min(k2, k3));
    }
    const int minS = (int)(key >> 16), best = (int)(key & 0xffffu);
    const int wgt = 100 - g.uniq, thr = minS * 100;
    bool reject;
    // S[best -+ 1] for the sub-pixel step (clamped: k_select uses them only for 0 < best < D-1)
    const int dm = max(best - 1, 0), dp = min(best + 1, D - 1);
    uint16_t *rw = reinterpret_cast<uint16_t *>(rows + lane * stride);
    const uint32_t nb = (uint32_t)rw[dm] | ((uint32_t)rw[dp] << 16);
    if (POSW) {
        // wgt > 0: one comparison against the smallest S outside best-1..best+1.  The row in LDS is
        // this lane's alone and not needed again: overwrite those three entries with MAX_COST and
        // take a plain packed minimum of the row.
        rw[dm] = (uint16_t)SGM_MAX_COST;
        rw[best] = (uint16_t)SGM_MAX_COST;
        rw[dp] = (uint16_t)SGM_MAX_COST;
        uint32_t far = SGM_SENT;
#pragma unroll 64
        for (int d0 = 0; d0 < D; d0 += 4) {
            const uint2 v = *reinterpret_cast<const uint2 *>(row + d0 * 2);
            far = pk_min_s(far, pk_min_s(v.x, v.y));
        }
        reject = (int)min(far & 0xffffu, far >> 16) * wgt < thr;
    } else {
        reject = false;
        for (int d = 0; d < D; d++) {
            const int sv = *reinterpret_cast<const uint16_t *>(row + d * 2);
            reject |= (sv * wgt < thr) && (abs(best - d) > 1);
        }
    }
    reject = reject || (minS == SGM_MAX_COST);
    const int64_t p = p0 + lane;
    const int y = (int)(p / W1), x = (int)(p - (int64_t)y * W1);
    wta[(int64_t)y * g.W + g.minX1 + x] = make_uint2(reject ? 0xffffffffu : key, nb);
    }  // lane < np
    }  // blocks of this workgroup
}

__device__ __forceinline__ void cswap(int &a, int &b)
{
    const int lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

// dst2 (nullable): a second copy of the result -- the engine's output buffer, on which the speckle
// filter then works in place, while dst stays readable as the "after median" tap
__global__ __launch_bounds__(256) void k_median3(const int16_t *__restrict__ src,
                                                 int16_t *__restrict__ dst, int16_t *__restrict__ dst2, int H, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int16_t *r0 = src + (int64_t)max(y - 1, 0) * W;
    const int16_t *r1 = src + (int64_t)y * W;
    const int16_t *r2 = src + (int64_t)min(y + 1, H - 1) * W;
    const int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    int p0 = r0[xl], p1 = r0[x], p2 = r0[xr], p3 = r1[xl], p4 = r1[x], p5 = r1[xr], p6 = r2[xl], p7 = r2[x], p8 = r2[xr];
    // 19-exchange median-of-9 network
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1); cswap(p3, p4); cswap(p6, p7);
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
    cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7); cswap(p4, p2); cswap(p6, p4);
    cswap(p4, p2);
    dst[(int64_t)y * W + x] = (int16_t)p4;
    if (dst2) dst2[(int64_t)y * W + x] = (int16_t)p4;
}

// ------------------------------------------------------------------------------------------
// Speckle filter (A.8) as connected components over horizontal RUNS: a run is a maximal stretch
// of one row whose neighbouring pixels are linked (both != newVal, |a-b| <= maxDiff).  Every
// pixel of a run points at the run's first pixel; first pixels form a lock-free union-find
// forest (links only ever decrease, atomicMin) joined where runs of adjacent rows touch.  The
// component partition, hence the output, does not depend on the traversal order upstream uses.
//
//   label[i] : -1 invalid | i's run start (non-start pixels, immutable) | parent link (starts)
//   rlen[s]  : length of the run starting at s          csz[r] : pixels of the component rooted at r
__device__ __forceinline__ int uf_load(const int *L, int a)
{
    // agent scope: a CU's L1 is never refreshed by other CUs' atomics
    return __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(int *L, int a)
{
    int p = uf_load(L, a);
    while (p != a) {
        const int gp = uf_load(L, p);
        if (gp != p) atomicMin(&L[a], gp);  // path halving; any ancestor is a valid parent
        a = p;
        p = gp;
    }
    return a;
}
__device__ __forceinline__ void uf_union(int *L, int a, int b)
{
    bool done;
    do {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) {
            const int old = atomicMin(&L[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&L[a], b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    } while (!done);
}

__device__ __forceinline__ bool linked(int a, int b, int newVal, int maxDiff)
{
    return a != newVal && b != newVal && abs(a - b) <= maxDiff;
}

// one wave per row: run starts by an inclusive prefix-max of "x where a run begins"
__global__ __launch_bounds__(64) void k_ccl_rows(const int16_t *__restrict__ img, int *__restrict__ label,
                                                 int *__restrict__ rlen, int *__restrict__ csz, int W,
                                                 int newVal, int maxDiff)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const int16_t *row = img + (int64_t)y * W;
    const int64_t base = (int64_t)y * W;
    int carry = -1;  // start of the run the previous chunk ended in
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool in = x < W;
        const int v = in ? row[x] : newVal;
        const int vl = (in && x > 0) ? row[x - 1] : newVal;
        const int vr = (in && x < W - 1) ? row[x + 1] : newVal;
        const bool valid = v != newVal;
        const bool starts = valid && !linked(v, vl, newVal, maxDiff);
        int s = starts ? x : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(s, o);
            if (lane >= o) s = max(s, t);
        }
        s = max(s, carry);
        if (in) {
            label[base + x] = valid ? (int)(base + s) : -1;
            csz[base + x] = 0;
            if (valid && !linked(v, vr, newVal, maxDiff)) rlen[base + s] = x - s + 1;  // run ends here
        }
        carry = __shfl(s, 63);
    }
}

// join runs of adjacent rows; one union per maximal stretch of vertical links between two runs
__global__ __launch_bounds__(256) void k_ccl_merge(const int16_t *__restrict__ img, int *label, int H, int W,
                                                   int newVal, int maxDiff)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y == 0) return;
    const int64_t i = (int64_t)y * W + x;
    const int v = img[i], u = img[i - W];
    if (!linked(v, u, newVal, maxDiff)) return;
    bool need = x == 0;
    if (!need) {
        const int vl = img[i - 1], ul = img[i - W - 1];
        // the stretch starts here if the left column is not linked the same way
        need = !linked(v, vl, newVal, maxDiff) || !linked(u, ul, newVal, maxDiff) || !linked(vl, ul, newVal, maxDiff);
    }
    if (!need) return;
    const bool vs = x == 0 || !linked(v, img[i - 1], newVal, maxDiff);
    const bool us = x == 0 || !linked(u, img[i - W - 1], newVal, maxDiff);
    const int a = vs ? (int)i : label[i];          // run starts (non-start labels are immutable)
    const int b = us ? (int)(i - W) : label[i - W];
    uf_union(label, a, b);
}

// per run: add its length to its component's root and point the run start at the root
__global__ __launch_bounds__(256) void k_ccl_count(const int16_t *__restrict__ img, int *label,
                                                   const int *__restrict__ rlen, int *__restrict__ csz, int H, int W,
                                                   int newVal, int maxDiff)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int64_t i = (int64_t)y * W + x;
    const int v = img[i];
    if (v == newVal) return;
    if (x > 0 && linked(v, img[i - 1], newVal, maxDiff)) return;  // not a run start
    const int r = uf_find(label, (int)i);
    atomicAdd(&csz[r], rlen[i]);
    if (r != (int)i) atomicMin(&label[i], r);  // run start -> root directly (all unions are done): k_ccl_apply needs one hop
}

__global__ __launch_bounds__(256) void k_ccl_apply(int16_t *__restrict__ img, int *label,
                                                   const int *__restrict__ csz, int H, int W, int newVal,
                                                   int maxDiff, int maxSpeckleSize)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int64_t i = (int64_t)y * W + x;
    const int l = label[i];
    if (l < 0) return;
    const int r = label[l];  // l is i itself or i's run start, which k_ccl_count pointed at the root
    if (csz[r] <= maxSpeckleSize) img[i] = (int16_t)newVal;
}

// ------------------------------------------------------------------------------------------
// main.ipynb:668-670 : f = i16 / 16 ; f * float(f > 0)   (invalid -> -0.0, zero -> +0.0)
__global__ __launch_bounds__(256) void k_disp_to_float(const int16_t *__restrict__ d, float *__restrict__ out,
                                                       int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = (float)d[i] / 16.0f;
    out[i] = f * (f > 0.0f ? 1.0f : 0.0f);
}

// order-preserving float <-> uint key for atomicMin
__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void k_min_f32(const float *__restrict__ d, int64_t n, uint32_t *minkey)
{
    uint32_t k = 0xffffffffu;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        k = min(k, fkey(d[i]));
    k = wave_min_u32(k);
    if ((threadIdx.x & 63) == 0) atomicMin(minkey, k);
}

struct QMat {
    double q[16];
};

// Appendix B: homogeneous point in double, sums in index order from 0, one rounding to float,
// divide as multiply by the reciprocal, one more rounding.  The library is built with
// -ffp-contract=off so that no fused multiply-add changes the double arithmetic.
__device__ __forceinline__ void reproject_px(float fd, int x, int y, int64_t i, const QMat &Q, const uint32_t *minkey,
                                             float *__restrict__ xyz)
{
    const double d = (double)fd;
    const double v[4] = {(double)x, (double)y, d, 1.0};
    double h[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) s += Q.q[r * 4 + k] * v[k];
        h[r] = s;
    }
    const double ia = 1.0 / h[3];
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float f = (float)h[r];
        o[r] = (float)((double)f * ia);
    }
    if (minkey) {
        const double mn = (double)fkey_inv(*minkey);
        if (fabs(d - mn) <= (double)FLT_EPSILON) o[2] = 10000.f;
    }
    xyz[i * 3 + 0] = o[0];
    xyz[i * 3 + 1] = o[1];
    xyz[i * 3 + 2] = o[2];
}

__global__ __launch_bounds__(256) void k_reproject(const float *__restrict__ disp, int H, int W, QMat Q,
                                                   const uint32_t *minkey, float *__restrict__ xyz)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int64_t i = (int64_t)y * W + x;
    reproject_px(disp[i], x, y, i, Q, minkey, xyz);
}

// The driver cell in one launch (main.ipynb:668-670 then :697): int16 disparity -> float disparity
// (stored if dispf is non-null) -> XYZ.  Same arithmetic as k_disp_to_float followed by k_reproject.
__global__ __launch_bounds__(256) void k_float_xyz(const int16_t *__restrict__ d16, int H, int W, QMat Q,
                                                   float *__restrict__ dispf, float *__restrict__ xyz)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int64_t i = (int64_t)y * W + x;
    const float f0 = (float)d16[i] / 16.0f;
    const float f = f0 * (f0 > 0.0f ? 1.0f : 0.0f);
    if (dispf) dispf[i] = f;
    reproject_px(f, x, y, i, Q, nullptr, xyz);
}

// main.ipynb:726-730
__global__ __launch_bounds__(256) void k_valid_mask(const float *__restrict__ xyz, const float *__restrict__ disp,
                                                    int64_t n, uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float X = xyz[i * 3];
    mask[i] = (uint8_t)(!isnan(X) && !isinf(X) && disp[i] > 0.0f);
}

// ------------------------------------------------------------------------------------------
// Valid-point compaction + colour gather (/root/reference/main.ipynb:726-737):
//   mask = ~isnan(X) & ~isinf(X) & (disp > 0);  valid_points = points_3D[mask];  valid_colors = colors[mask]
// numpy boolean indexing keeps row-major order, so the compaction is an ordered one:
// per-256-pixel block counts -> exclusive scan -> ordered scatter (ballot + popcount inside a wave).
__device__ __forceinline__ bool point_valid(const float *xyz, const float *disp, int64_t i)
{
    const float X = xyz[i * 3];
    return !isnan(X) && !isinf(X) && disp[i] > 0.0f;
}

__global__ __launch_bounds__(256) void k_compact_count(const float *__restrict__ xyz, const float *__restrict__ disp,
                                                       int64_t n, uint32_t *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool v = i < n && point_valid(xyz, disp, i);
    const unsigned long long b = __builtin_amdgcn_ballot_w64(v);
    __shared__ uint32_t wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of `m` block counts by one workgroup (m is a few 10^4 at most); total -> counts[m]
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t *counts, int m)
{
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int per = (m + 1023) / 1024;
    const int lo = min(t * per, m), hi = min(lo + per, m);
    uint32_t s = 0;
    for (int i = lo; i < hi; i++) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan of the partials
        const uint32_t v = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0u;
    for (int i = lo; i < hi; i++) {
        const uint32_t c = counts[i];
        counts[i] = run;
        run += c;
    }
    if (t == 1023) counts[m] = part[1023];
}

__global__ __launch_bounds__(256) void k_compact_scatter(const float *__restrict__ xyz, const float *__restrict__ disp,
                                                         const uint8_t *__restrict__ colors, int64_t n,
                                                         const uint32_t *__restrict__ offsets,
                                                         float *__restrict__ out_xyz, uint8_t *__restrict__ out_rgb)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool v = i < n && point_valid(xyz, disp, i);
    const unsigned long long b = __builtin_amdgcn_ballot_w64(v);
    __shared__ uint32_t wsum[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wsum[w] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t base = offsets[blockIdx.x];
    for (int k = 0; k < w; k++) base += wsum[k];
    if (v) {
        const uint32_t o = base + (uint32_t)__builtin_popcountll(b & ((1ull << lane) - 1ull));
        out_xyz[(int64_t)o * 3 + 0] = xyz[i * 3 + 0];
        out_xyz[(int64_t)o * 3 + 1] = xyz[i * 3 + 1];
        out_xyz[(int64_t)o * 3 + 2] = xyz[i * 3 + 2];
        if (colors) {
            out_rgb[(int64_t)o * 3 + 0] = colors[i * 3 + 0];
            out_rgb[(int64_t)o * 3 + 1] = colors[i * 3 + 1];
            out_rgb[(int64_t)o * 3 + 2] = colors[i * 3 + 2];
        }
    }
}

// the total of a compaction (counts[m] after k_compact_scan) as the int64 the caller's device memory holds
__global__ void k_compact_total(const uint32_t *__restrict__ total, int64_t *__restrict__ out) { *out = (int64_t)*total; }

}  // namespace sgm

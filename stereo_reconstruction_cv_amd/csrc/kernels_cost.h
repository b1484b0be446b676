// kernels_cost.h -- matching-cost stage: prefilter + Birchfield-Tomasi pixel cost + box sum.
//
// Replaces upstream calcPixelCostBT / hsumBuf / C update inside cv2.StereoSGBM.compute
// (/root/reference/main.ipynb:668; arithmetic restated in SURVEY.md A.2-A.4 and
// oracle/sgbm_oracle.c: row_features, bt_row, hsum_row, the C update in sgbm_core).
//
//   k_features : u8 image -> per-pixel (value, interval lo, interval hi) for the gradient
//                and raw channels.  Left image: one packed 8-byte record per pixel (read
//                wave-uniformly).  Right image: six byte planes stored MIRRORED in x so that
//                ascending disparity is ascending address.
//   k_hsum     : one wave per (row, column chunk); lanes span the disparities, the wave walks
//                x keeping a sliding window of right-image features in registers and a ring of
//                the last blockSize+1 pixel-cost vectors in LDS -> horizontal running box sum.
//   k_vsum     : vertical running box sum of hsum rows -> block cost C (no +P2 bias).
#pragma once
#include "sgm_device.h"

namespace sgm {

struct Geom {
    int H, W;
    int minD, D;
    int minX1, W1;
    int SW2, SH2;
    int P1, P2;
    int uniq, d12;
    int ftzero;
    int invalid_scaled;
    int mode;
    int NP;       // packed pairs per lane
    int64_t rowsz;  // W1 * D
    uint32_t *hr;   // headroom record of this compute (sgm_device.h: headroom_commit_pk), may be null
};

// ------------------------------------------------------------------------------------------
// blockIdx.z = 0: left image -> left_rec; 1: right image -> right_planes (one launch for the pair)
__global__ __launch_bounds__(256) void k_features(const uint8_t *__restrict__ imgL, const uint8_t *__restrict__ imgR,
                                                  int64_t stride, int H, int W, int ftzero,
                                                  uint2 *__restrict__ left_rec_,
                                                  uint8_t *__restrict__ right_planes_)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const bool is_right = blockIdx.z != 0;
    const uint8_t *img = is_right ? imgR : imgL;
    uint2 *left_rec = is_right ? nullptr : left_rec_;
    uint8_t *right_planes = is_right ? right_planes_ : nullptr;
    const uint8_t *row = img + (int64_t)y * stride;
    const uint8_t *up = y > 0 ? row - stride : row;
    const uint8_t *dn = y < H - 1 ? row + stride : row;

    int pf[3], rw[3];  // values at x-1, x, x+1 (only read where they exist)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int xx = x + k - 1;
        if (xx <= 0 || xx >= W - 1) {  // border columns hold ftzero in BOTH channels (A.2)
            pf[k] = ftzero;
            rw[k] = ftzero;
        } else {
            int g = 2 * ((int)row[xx + 1] - (int)row[xx - 1]) + ((int)up[xx + 1] - (int)up[xx - 1]) +
                    ((int)dn[xx + 1] - (int)dn[xx - 1]);
            pf[k] = min(max(g, -ftzero), ftzero) + ftzero;
            rw[k] = row[xx];
        }
    }
    uint32_t out[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int *v = c == 0 ? pf : rw;
        const int a = v[1];
        const int l = x > 0 ? (a + v[0]) / 2 : a;
        const int r = x < W - 1 ? (a + v[2]) / 2 : a;
        const int lo = min(a, min(l, r)), hi = max(a, max(l, r));
        out[c] = (uint32_t)a | ((uint32_t)lo << 8) | ((uint32_t)hi << 16);
    }
    if (left_rec) left_rec[(int64_t)y * W + x] = make_uint2(out[0], out[1]);
    if (right_planes) {
        const int64_t psz = (int64_t)H * W;
        const int64_t o = (int64_t)y * W + (W - 1 - x);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            right_planes[(c * 3 + 0) * psz + o] = (uint8_t)(out[c] & 0xff);
            right_planes[(c * 3 + 1) * psz + o] = (uint8_t)((out[c] >> 8) & 0xff);
            right_planes[(c * 3 + 2) * psz + o] = (uint8_t)((out[c] >> 16) & 0xff);
        }
    }
}

// Birchfield-Tomasi on packed pairs: min( dist(u, [v0,v1]), dist(v, [u0,u1]) )   (A.3)
__device__ __forceinline__ uint32_t bt_pair(uint32_t U, uint32_t U0, uint32_t U1, uint32_t V,
                                            uint32_t V0, uint32_t V1)
{
    uint32_t c0 = pk_max_u(pk_subs_u(U, V1), pk_subs_u(V0, U));
    uint32_t c1 = pk_max_u(pk_subs_u(V, U1), pk_subs_u(U0, V));
    return pk_min_u(c0, c1);
}

// LDS layout of one k_hsum workgroup (one wave): [ring RS*64*NP dwords][left records][6 planes]
struct HsumLds {
    int ring_bytes, lrec_bytes, seg_len, total_bytes;
};
static inline HsumLds hsum_lds_layout(int NP, int RS, int XL, int SW2)
{
    HsumLds l;
    l.ring_bytes = RS * 64 * NP * 4;
    int nj = XL + 2 * SW2 + 2;
    l.lrec_bytes = ((nj * 8) + 15) & ~15;
    l.seg_len = (nj + 128 * NP + 15) & ~15;
    l.total_bytes = l.ring_bytes + l.lrec_bytes + 6 * l.seg_len;
    return l;
}

// byte B of x in both halves of a packed pair (one v_perm_b32)
template <int B> __device__ __forceinline__ uint32_t splat_byte(uint32_t x)
{
    return __builtin_amdgcn_perm(x, x, 0x0c000c00u | (uint32_t)B | ((uint32_t)B << 16));
}

// One wave per (row, chunk of XL output columns).  Column j of the pixel cost is computed once (BT
// of the left record against the sliding windows of the six right-image planes), kept in an LDS
// ring of RS columns, and the running horizontal sum hs(x) = hs(x-1) + pix(x+r) - pix(x-r-1)
// (A.4, clamped at the domain edges) is stored as soon as its right-most tap exists.
//
// RS_T > 0 (= RS, a power of two) adds the interior fast path: RS_T columns per iteration,
// unrolled, so ring slots, window taps and record reads are immediate offsets and the only
// scalar work per column is the store offset; it covers every column whose window is not
// clamped.  The generic per-column step handles the first and last columns of a chunk / row.
template <int NP, int RS_T>
__global__ __launch_bounds__(64) void k_hsum(Geom g, const uint2 *__restrict__ lrec,
                                             const uint8_t *__restrict__ rplanes,
                                             int16_t *__restrict__ hsum, int XL, int nchunks, int RS,
                                             int ring_bytes, int lrec_bytes, int seg_len, int y_base)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *ring = reinterpret_cast<uint32_t *>(smem);
    uint2 *lds_lrec = reinterpret_cast<uint2 *>(smem + ring_bytes);
    uint8_t *seg = smem + ring_bytes + lrec_bytes;

    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int yr = __builtin_amdgcn_readfirstlane(unit / nchunks), ck = unit - yr * nchunks;  // (uniform: see uniform_rsrc)
    const int y = y_base + yr;  // the launch covers rows y_base .. y_base + gridDim.x / nchunks - 1
    const int W1 = g.W1, SW2 = g.SW2, W = g.W;
    const int xs = ck * XL, xe = min(xs + XL, W1);
    const int j0 = max(xs - SW2 - 1, 0), j1 = min(xe - 1 + SW2, W1 - 1);
    const int nj = j1 - j0 + 1;
    const bool active = 2 * NP * lane < g.D;

    // ---- stage this row's features for the chunk ----
    for (int k = lane; k < nj; k += 64) lds_lrec[k] = lrec[(int64_t)y * W + (j0 + k + g.minX1)];
    {
        // mirrored position of (column j, disparity index e): (W-1-(j+minX1)) + minD + e
        const int base_j1 = W - 1 - (j1 + g.minX1) + g.minD;
        const int len = (j1 - j0) + 128 * NP;
        const int64_t psz = (int64_t)g.H * W;
        for (int s = lane; s < len; s += 64) {
            const int pos = base_j1 + s;
            const bool ok = pos >= 0 && pos < W;
#pragma unroll
            for (int c = 0; c < 6; c++)
                seg[c * seg_len + s] = ok ? rplanes[c * psz + (int64_t)y * W + pos] : (uint8_t)0;
        }
    }
    __syncthreads();  // single wave; orders the LDS staging before the reads below

    // ---- sliding windows of the six right-image planes ----
    uint32_t w[6][NP];
    {
        const int off = (j1 - j0) + 2 * NP * lane;
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int i = 0; i < NP; i++)
                w[c][i] = (uint32_t)seg[c * seg_len + off + 2 * i] | ((uint32_t)seg[c * seg_len + off + 2 * i + 1] << 16);
    }

    uint32_t hs[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) hs[i] = 0;
    int next_x = xs;
    // this row of the output as a buffer resource; lanes past D get an out-of-range offset, so
    // their stores are dropped by the bounds check instead of by a branch
    const int row_bytes = W1 * g.D * 2;
    const __amdgpu_buffer_rsrc_t orow = uniform_rsrc(hsum, (int64_t)y * g.rowsz * 2, row_bytes);
    const int voff = active ? 4 * NP * lane : row_bytes;
    const int pxb = g.D * 2;

    // the generic step: column j with every clamp and the first-output special case
    auto column = [&](int j) {
        if (j > j0) {
            const int off = (j1 - j) + 2 * NP * lane;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const uint32_t nw = seg[c * seg_len + off];
#pragma unroll
                for (int i = NP - 1; i >= 1; i--) w[c][i] = __builtin_amdgcn_alignbit(w[c][i], w[c][i - 1], 16);
                w[c][0] = (w[c][0] << 16) | nw;
            }
        }
        const uint2 rec = lds_lrec[j - j0];
        const uint32_t U = splat_byte<0>(rec.x), U0 = splat_byte<1>(rec.x), U1 = splat_byte<2>(rec.x);
        const uint32_t R = splat_byte<0>(rec.y), R0 = splat_byte<1>(rec.y), R1 = splat_byte<2>(rec.y);
        uint32_t *slot = ring + ((j & (RS - 1)) * 64 + lane) * NP;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            uint32_t a = bt_pair(U, U0, U1, w[0][i], w[1][i], w[2][i]);
            uint32_t b = bt_pair(R, R0, R1, w[3][i], w[4][i], w[5][i]);
            slot[i] = pk_add(a, pk_shr_u(b, 2));
        }
        // every output column whose right-most (clamped) tap is now available
        while (next_x < xe && min(next_x + SW2, W1 - 1) <= j) {
            const int x = next_x;
            if (x == xs) {
#pragma unroll
                for (int i = 0; i < NP; i++) hs[i] = 0;
                for (int t = -SW2; t <= SW2; t++) {
                    const int jc = min(max(x + t, 0), W1 - 1);
                    const uint32_t *p = ring + ((jc & (RS - 1)) * 64 + lane) * NP;
#pragma unroll
                    for (int i = 0; i < NP; i++) hs[i] = pk_add(hs[i], p[i]);
                }
            } else {
                const uint32_t *pa = ring + ((min(x + SW2, W1 - 1) & (RS - 1)) * 64 + lane) * NP;
                const uint32_t *pb = ring + ((max(x - SW2 - 1, 0) & (RS - 1)) * 64 + lane) * NP;
#pragma unroll
                for (int i = 0; i < NP; i++) hs[i] = pk_sub(pk_add(hs[i], pa[i]), pb[i]);
            }
            Pack<NP> o;
#pragma unroll
            for (int i = 0; i < NP; i++) o.r[i] = hs[i];
            buf_store<NP>(o, orow, voff, x * pxb);
            next_x++;
        }
    };

    int j = j0;
    if (RS_T > 0) {
        const int bs = 2 * SW2 + 1;
        // fast columns: exactly one output x = j - SW2, not the chunk's first, no clamped tap:
        //   x > xs, x - SW2 - 1 >= max(0, j0) (the outgoing tap is in the ring), j <= W1 - 2
        int ja = max(xs + 1 + SW2, j0 + bs);
        ja = (ja + RS_T - 1) & ~(RS_T - 1);
        const int jb = min(xe - 1 + SW2, W1 - 2);  // last fast column
        if (ja + RS_T - 1 <= jb) {
            for (; j < ja; j++) column(j);
            const uint32_t *ring_lane = ring + lane * NP;
            uint32_t *ring_lane_w = ring + lane * NP;
            // per-lane tap addresses of the six planes for the block's LAST column (offsets then
            // count up towards the block's first column), and the record address of its first
            int tap[6];
#pragma unroll
            for (int c = 0; c < 6; c++) tap[c] = c * seg_len + 2 * NP * lane + (j1 - j) - (RS_T - 1);
            // the record address is the same in every lane; hidden from the compiler so that the
            // record stays in VGPRs (v_perm splats) instead of a readfirstlane + scalar unpack
            int recp = j - j0;
            asm volatile("" : "+v"(recp));
            int so = (j - SW2) * pxb;
            for (; j + RS_T - 1 <= jb; j += RS_T) {
#pragma unroll
                for (int u = 0; u < RS_T; u++) {  // column j + u, ring slot u
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        const uint32_t nw = seg[tap[c] + (RS_T - 1 - u)];
#pragma unroll
                        for (int i = NP - 1; i >= 1; i--) w[c][i] = __builtin_amdgcn_alignbit(w[c][i], w[c][i - 1], 16);
                        w[c][0] = (w[c][0] << 16) | nw;
                    }
                    const uint2 rec = lds_lrec[recp + u];
                    const uint32_t U = splat_byte<0>(rec.x), U0 = splat_byte<1>(rec.x), U1 = splat_byte<2>(rec.x);
                    const uint32_t R = splat_byte<0>(rec.y), R0 = splat_byte<1>(rec.y), R1 = splat_byte<2>(rec.y);
                    const uint32_t *old = ring_lane + ((u - bs) & (RS_T - 1)) * 64 * NP;  // column j + u - bs
                    Pack<NP> o;
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        const uint32_t a = bt_pair(U, U0, U1, w[0][i], w[1][i], w[2][i]);
                        const uint32_t b = bt_pair(R, R0, R1, w[3][i], w[4][i], w[5][i]);
                        const uint32_t pix = pk_add(a, pk_shr_u(b, 2));
                        hs[i] = pk_sub(pk_add(hs[i], pix), old[i]);
                        ring_lane_w[u * 64 * NP + i] = pix;
                        o.r[i] = hs[i];
                    }
                    buf_store<NP>(o, orow, voff, so);
                    so += pxb;
                }
#pragma unroll
                for (int c = 0; c < 6; c++) tap[c] -= RS_T;
                recp += RS_T;
            }
            next_x = j - SW2;
        }
    }
    for (; j <= j1; j++) column(j);
}

// ---- byte pipeline of the block cost (default for window radii 1..5) ----------------------------
// The per-pixel Birchfield-Tomasi cost is at most 2*ftzero + 63 <= 255: stored as one BYTE per
// (x, d) it is half the volume of the int16 horizontal sums that k_hsum hands to k_vsum, and the
// box filter becomes one kernel that reads it back (k_box_u8).  HBM traffic of the cost stage:
// 0.5 V written + ~0.55 V read + V written, against 3.1 V of k_hsum + k_vsum_ring.

// pix(x, y, d) = BT(gradient channel) + (BT(raw channel) >> 2), one wave per (row, chunk of XL
// columns), lanes span the disparities, sliding right-feature windows as in k_hsum.
template <int NP>
__global__ __launch_bounds__(64) void k_pix(Geom g, const uint2 *__restrict__ lrec, const uint8_t *__restrict__ rplanes,
                                            uint8_t *__restrict__ pix, int XL, int nchunks, int lrec_bytes, int seg_len,
                                            int y_base /* the launch covers rows y_base .. y_base + gridDim.x / nchunks - 1 */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *seg = smem + lrec_bytes;  // (the first lrec_bytes are unused since the left records are read by scalar loads)
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int yr = __builtin_amdgcn_readfirstlane(unit / nchunks), ck = unit - yr * nchunks;  // (uniform: see uniform_rsrc)
    const int y = y_base + yr;
    const int W1 = g.W1, W = g.W;
    const int j0 = ck * XL, j1 = min(j0 + XL, W1) - 1;
    const bool active = 2 * NP * lane < g.D;
    {
        const int base_j1 = W - 1 - (j1 + g.minX1) + g.minD;  // mirrored position of (column j1, disparity index 0)
        const int len = (j1 - j0) + 128 * NP;
        const int64_t psz = (int64_t)g.H * W;
        for (int s = lane; s < len; s += 64) {
            const int pos = base_j1 + s;
            const bool ok = pos >= 0 && pos < W;
#pragma unroll
            for (int c = 0; c < 6; c++)
                seg[c * seg_len + s] = ok ? rplanes[c * psz + (int64_t)y * W + pos] : (uint8_t)0;
        }
    }
    __syncthreads();
    uint32_t w[6][NP];
    {
        const int off = (j1 - j0) + 2 * NP * lane;
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int i = 0; i < NP; i++)
                w[c][i] = (uint32_t)seg[c * seg_len + off + 2 * i] | ((uint32_t)seg[c * seg_len + off + 2 * i + 1] << 16);
    }
    // this row of the output; lanes past D store nowhere (offset beyond the descriptor)
    const int row_bytes = W1 * g.D;
    const __amdgpu_buffer_rsrc_t orow = uniform_rsrc(pix, (int64_t)y * row_bytes, row_bytes);
    const int voff = active ? 2 * NP * lane : row_bytes;
    // The left pixel's record is the same for every lane: it is read with SCALAR loads straight from `lrec` and its six
    // bytes are splat into packed pairs on the scalar unit -- these launches run several waves per SIMD, so scalar
    // work rides beside the other waves' vector instructions, and the kernel is bound by the vector ALU (round 3:
    // 58 -> 47 vector instructions per column at NP = 2; the six v_perm splats of a VGPR-held record and the LDS tap
    // address additions -- now immediates on a per-iteration base -- were a fifth of them).
    // (constant address space: k_features wrote the records before this launch; hipcc uses scalar loads only there -- through
    // the global pointer the buffer stores below count as possible writers and the loads stay vector loads)
    typedef const __attribute__((address_space(4))) v2u32 *crec_ptr;  // (a plain vector type: HIP's uint2 class has no constructor from that address space)
    // (wave-uniform address; v_readfirstlane on both halves, or the 64-bit row arithmetic keeps it in VGPRs: see uniform_rsrc)
    const uint64_t la = (uint64_t)(uintptr_t)(lrec + ((int64_t)y * W + (j0 + g.minX1)));
    const uint32_t la_lo = __builtin_amdgcn_readfirstlane((uint32_t)la), la_hi = __builtin_amdgcn_readfirstlane((uint32_t)(la >> 32));
    const crec_ptr lrow = (crec_ptr)(uintptr_t)(((uint64_t)la_hi << 32) | la_lo);  // (uint32_t first: the builtin returns int)
    auto splat = [](uint32_t v) __attribute__((always_inline)) { return v | (v << 16); };
    int so = j0 * g.D;
    // one column: shift the six windows by the new bytes nw[] (not for the chunk's first column), cost, store
    auto column = [&](const v2u32 rec, bool shift, const uint32_t (&nw)[6]) __attribute__((always_inline)) {
        if (shift) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
#pragma unroll
                for (int i = NP - 1; i >= 1; i--) w[c][i] = __builtin_amdgcn_alignbit(w[c][i], w[c][i - 1], 16);
                w[c][0] = (w[c][0] << 16) | nw[c];
            }
        }
        const uint32_t U = splat(rec.x & 0xffu), U0 = splat((rec.x >> 8) & 0xffu), U1 = splat((rec.x >> 16) & 0xffu);
        const uint32_t R = splat(rec.y & 0xffu), R0 = splat((rec.y >> 8) & 0xffu), R1 = splat((rec.y >> 16) & 0xffu);
        uint32_t pv[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t a = bt_pair(U, U0, U1, w[0][i], w[1][i], w[2][i]);
            const uint32_t b = bt_pair(R, R0, R1, w[3][i], w[4][i], w[5][i]);
            pv[i] = pk_add(a, pk_shr_u(b, 2));
        }
        // low byte of every int16 half: 2*NP bytes per lane, ascending d
        if constexpr (NP == 1) {
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)__builtin_amdgcn_perm(0u, pv[0], 0x0c0c0200u), orow, voff, so, 0);
        } else if constexpr (NP == 2) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(pv[1], pv[0], 0x06040200u), orow, voff, so, 0);
        } else {
            v2u32 o;
            o.x = __builtin_amdgcn_perm(pv[1], pv[0], 0x06040200u);
            o.y = __builtin_amdgcn_perm(pv[3], pv[2], 0x06040200u);
            __builtin_amdgcn_raw_buffer_store_b64(o, orow, voff, so, 0);
        }
        so += g.D;
    };
    const uint32_t none[6] = {0, 0, 0, 0, 0, 0};
    column(lrow[0], false, none);
    int j = j0 + 1;
    // per-lane LDS address of the six planes' new byte for column j + 3 (the last of a block of four); the bytes of
    // columns j .. j + 3 are then at immediate offsets 3 .. 0
    int tp[6];
#pragma unroll
    for (int c = 0; c < 6; c++) tp[c] = c * seg_len + 2 * NP * lane + (j1 - j0) - (j - j0) - 3;
    // four columns per iteration: their 24 window bytes are read first (one counted wait instead of a
    // full lgkmcnt(0) drain per column), then the four columns are computed from registers
    for (; j + 3 <= j1; j += 4) {
        uint32_t nb[4][6];
        v2u32 rec[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rec[u] = lrow[j - j0 + u];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < 6; c++) nb[u][c] = seg[tp[c] + (3 - u)];
#pragma unroll
        for (int c = 0; c < 6; c++) tp[c] -= 4;
#pragma unroll
        for (int u = 0; u < 4; u++) column(rec[u], true, nb[u]);
    }
    for (; j <= j1; j++) {
        uint32_t nb[6];
#pragma unroll
        for (int c = 0; c < 6; c++) nb[c] = seg[tp[c] + 3];
#pragma unroll
        for (int c = 0; c < 6; c++) tp[c] -= 1;
        column(lrow[j - j0], true, nb);
    }
}

// C(x, y, d) = sum over the (2R+1)^2 window of pix, window clamped at the edges of the valid-column
// domain and of the image (A.4).  One wave per XC = 4 adjacent columns, lanes span the disparities,
// the wave walks down a band of rows.  Per row it loads the 2R+XC neighbouring columns' bytes
// (adjacent in memory), widens them once, forms hs of its first column by 2R additions and of the
// next three by hs(x+1) = hs(x) + col(x+R+1) - col(x-R) (true for clamped column indices as well);
// the vertical running sums keep the last 2R+2 hs vectors of each column in statically indexed
// register rings.  2.5 bytes read per output byte pair instead of 7: the first version (one column
// per wave) was bound by L2 traffic.
// GW < 64 (D <= 64, NP = 1): lane groups as in kernels_group.h -- a wave takes 64 / GW groups of XC columns, GW lanes
// span the disparities of a group (round 3: small D used the int16 pipeline, 3 V more traffic, because lanes spanning
// D = 16 would idle 7/8 of this kernel).
template <int R, int NP, int GW = 64>
__global__ __launch_bounds__(256) void k_box_u8(Geom g, const uint8_t *__restrict__ pix, int16_t *__restrict__ C, int RB)
{
    constexpr int RS = R <= 1 ? 4 : (R <= 3 ? 8 : 16);  // pow2 >= 2R+2
    constexpr int XC = 4, NT = 2 * R + XC;
    constexpr int NQ = (NP + 1) / 2;  // dwords of bytes per lane and column
    constexpr int G = 64 / GW;
    static_assert(GW == 64 || NP == 1, "lane groups: D <= 64");
    const int lane = GW == 64 ? (int)(threadIdx.x & 63) : (int)(threadIdx.x & 63) % GW;  // index along the disparities
    const int xw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * XC * G;                       // first column of the wave
    const int x0 = GW == 64 ? xw : xw + (int)((threadIdx.x & 63) / GW) * XC;             // ... of this lane's group
    const int W1 = g.W1, D = g.D, H = g.H;
    if (xw >= W1) return;  // (the whole wave; a GROUP past the row end loads clamped columns and stores nothing)
    const bool active = 2 * NP * lane < D;
    const int y0 = (int)blockIdx.y * RB, y1 = min(y0 + RB, H);
    const int row_bytes = W1 * D;
    // (the host takes this pipeline only while H * row_bytes < 2 GiB; plain 32-bit scalar arithmetic -- a
    // min<int64_t>() here went through v_min_f64 and dragged the descriptor into VGPRs: see uniform_rsrc)
    const __amdgpu_buffer_rsrc_t prs = uniform_rsrc(pix, 0, H * row_bytes);
    int coff[NT];  // byte offset of the clamped neighbour columns inside a row, this lane's disparities
#pragma unroll
    for (int t = 0; t < NT; t++) coff[t] = min(max(x0 + t - R, 0), W1 - 1) * D + 2 * NP * lane;
    // hs(x0 + c, y) for c = 0..XC-1: NP packed pairs each
    auto hsum_row = [&](int y, uint32_t (&hs)[XC][NP]) {
        const int so = min(max(y, 0), H - 1) * row_bytes;  // (< 2 GiB: larger pix volumes use the int16 pipeline)
        uint32_t v[NT][NQ];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if constexpr (NP == 1) v[t][0] = __builtin_amdgcn_raw_buffer_load_b16(prs, coff[t], so, 0);
            else if constexpr (NP == 2) v[t][0] = __builtin_amdgcn_raw_buffer_load_b32(prs, coff[t], so, 0);
            else {
                const v2u32 q = __builtin_amdgcn_raw_buffer_load_b64(prs, coff[t], so, 0);
                v[t][0] = q.x;
                v[t][1] = q.y;
            }
        }
        uint32_t u[NT][NP];  // widened: byte 2i, 2i+1 -> packed pair i
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < NP; i++)
                u[t][i] = (i & 1) ? __builtin_amdgcn_perm(0u, v[t][i / 2], 0x0c030c02u) : __builtin_amdgcn_perm(0u, v[t][i / 2], 0x0c010c00u);
#pragma unroll
        for (int i = 0; i < NP; i++) {
            uint32_t h = u[0][i];
#pragma unroll
            for (int t = 1; t <= 2 * R; t++) h = pk_add(h, u[t][i]);
            hs[0][i] = h;
#pragma unroll
            for (int c = 1; c < XC; c++) {
                h = pk_sub(pk_add(h, u[2 * R + c][i]), u[c - 1][i]);
                hs[c][i] = h;
            }
        }
    };
    uint32_t ring[RS][XC][NP], acc[XC][NP], hs[XC][NP];
    uint32_t hmax = 0;  // packed running maximum of every int16 lane upstream would hold (headroom record)
#pragma unroll
    for (int c = 0; c < XC; c++)
#pragma unroll
        for (int i = 0; i < NP; i++) acc[c][i] = 0;
    // window of row y0: rows y0-R .. y0+R (y0 % RS == 0: slot(y0 + j) = j & (RS-1) is static)
#pragma unroll
    for (int j = -R; j <= R; j++) {
        hsum_row(y0 + j, hs);
#pragma unroll
        for (int c = 0; c < XC; c++)
#pragma unroll
            for (int i = 0; i < NP; i++) {
                ring[j & (RS - 1)][c][i] = hs[c][i];
                acc[c][i] = pk_add(acc[c][i], hs[c][i]);
            }
    }
    const int64_t rowsz = (int64_t)W1 * D;
    int16_t *out = C + (int64_t)x0 * D + 2 * NP * lane;
    auto put = [&](int y) {
        if (active) {
#pragma unroll
            for (int c = 0; c < XC; c++)
                if (x0 + c < W1) {
                    Pack<NP> o;
#pragma unroll
                    for (int i = 0; i < NP; i++) o.r[i] = acc[c][i];
                    o.store(out + (int64_t)y * rowsz + c * D);
                }
        }
    };
    put(y0);
    // upstream reaches the first row of a band through C(y0-1) + hsum(y0+R) = C(y0) + hsum(y0-R-1)
    if (y0 > 0) hsum_row(y0 - R - 1, hs);
#pragma unroll
    for (int c = 0; c < XC; c++)
#pragma unroll
        for (int i = 0; i < NP; i++)
            if (x0 + c < W1) hmax = pk_max_u(hmax, y0 > 0 ? pk_add(acc[c][i], hs[c][i]) : acc[c][i]);
    for (int yb = y0; yb < y1; yb += RS) {
#pragma unroll
        for (int u = 0; u < RS; u++) {
            const int y = yb + u;
            if (y > y0 && y < y1) {
                hsum_row(y + R, hs);
#pragma unroll
                for (int c = 0; c < XC; c++)
#pragma unroll
                    for (int i = 0; i < NP; i++) {
                        const uint32_t t = pk_add(acc[c][i], hs[c][i]);  // C(y-1) + hsum(y+R): an int16 lane upstream
                        if (x0 + c < W1) hmax = pk_max_u(hmax, t);       // (columns past W1 hold no real sums)
                        acc[c][i] = pk_sub(t, ring[(u - R - 1) & (RS - 1)][c][i]);
                        ring[(u + R) & (RS - 1)][c][i] = hs[c][i];
                    }
                put(y);
            }
        }
    }
    if (!active) hmax = 0;
    headroom_commit_pk(g.hr, 0, hmax);
}

// ---- small D (the engine uses these for D <= 32): one THREAD per pixel ----------------------------
// With lanes spanning the disparities, D = 16 uses 8 of 64 lanes in k_hsum / k_pix.  Here a thread
// owns a pixel and loops over its disparities; the right-image features of a 256-pixel block
// (256 + D bytes per plane) are staged in LDS.  k_pix_px writes the per-pixel cost as bytes,
// k_hsum_px turns it into the int16 horizontal box sums that k_vsum_ring (element-wise, any D)
// finishes.  Same arithmetic as k_hsum (A.3, A.4).
__global__ __launch_bounds__(256) void k_pix_px(Geom g, const uint2 *__restrict__ lrec, const uint8_t *__restrict__ rplanes,
                                                uint8_t *__restrict__ pix)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t seg[];  // [6][256 + D]
    const int t = threadIdx.x, y = blockIdx.y, b0 = blockIdx.x * 256;
    const int W = g.W, W1 = g.W1, D = g.D, SL = 256 + D;
    // mirrored position of (column xi, disparity index e): (W-1-(xi+minX1)) + minD + e; the block's
    // smallest one belongs to xi = b0 + 255, e = 0
    const int pmin = W - 1 - (b0 + 255 + g.minX1) + g.minD;
    const int64_t psz = (int64_t)g.H * W;
    for (int s = t; s < SL - 1; s += 256) {
        const int pos = pmin + s;
        const bool ok = pos >= 0 && pos < W;
#pragma unroll
        for (int c = 0; c < 6; c++) seg[c * SL + s] = ok ? rplanes[c * psz + (int64_t)y * W + pos] : (uint8_t)0;
    }
    __syncthreads();
    const int xi = b0 + t;
    if (xi >= W1) return;
    const uint2 rec = lrec[(int64_t)y * W + xi + g.minX1];
    const int U = rec.x & 0xff, U0 = (rec.x >> 8) & 0xff, U1 = (rec.x >> 16) & 0xff;
    const int R = rec.y & 0xff, R0 = (rec.y >> 8) & 0xff, R1 = (rec.y >> 16) & 0xff;
    const uint8_t *sp = seg + (255 - t);
    uint32_t *out = reinterpret_cast<uint32_t *>(pix + ((int64_t)y * W1 + xi) * D);
    for (int e0 = 0; e0 < D; e0 += 4) {
        uint32_t packed = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = e0 + q;
            const int V = sp[0 * SL + e], V0 = sp[1 * SL + e], V1 = sp[2 * SL + e];
            const int Q = sp[3 * SL + e], Q0 = sp[4 * SL + e], Q1 = sp[5 * SL + e];
            const int a = min(max(max(U - V1, V0 - U), 0), max(max(V - U1, U0 - V), 0));
            const int b = min(max(max(R - Q1, Q0 - R), 0), max(max(Q - R1, R0 - Q), 0));
            packed |= (uint32_t)(a + (b >> 2)) << (8 * q);
        }
        out[e0 / 4] = packed;
    }
}

__global__ __launch_bounds__(256) void k_hsum_px(Geom g, const uint8_t *__restrict__ pix, int16_t *__restrict__ hsum)
{
    const int xi = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int W1 = g.W1, D = g.D, R = g.SW2;
    if (xi >= W1) return;
    const uint8_t *row = pix + (int64_t)y * W1 * D;
    uint32_t *out = reinterpret_cast<uint32_t *>(hsum + ((int64_t)y * W1 + xi) * D);
    for (int e0 = 0; e0 < D; e0 += 16) {  // 16 disparities = one 16-byte vector per neighbour
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int tt = -R; tt <= R; tt++) {
            const int xc = min(max(xi + tt, 0), W1 - 1);
            const uint4 v = *reinterpret_cast<const uint4 *>(row + (int64_t)xc * D + e0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc[2 * k] = pk_add(acc[2 * k], __builtin_amdgcn_perm(0u, w[k], 0x0c010c00u));
                acc[2 * k + 1] = pk_add(acc[2 * k + 1], __builtin_amdgcn_perm(0u, w[k], 0x0c030c02u));
            }
        }
        uint4 *o = reinterpret_cast<uint4 *>(out + e0 / 2);
        o[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        o[1] = make_uint4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// C(y) = sum_{j=-SH2..SH2} hsum(clamp(y+j, 0, H-1)), running along y inside a band of rows.
__global__ __launch_bounds__(256) void k_vsum(const int16_t *__restrict__ hs, int16_t *__restrict__ C,
                                              int H, int64_t rowsz, int SH2, int RB, uint32_t *hr)
{
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;  // 8 int16 per thread
    uint32_t hmax = 0;
    if (e < rowsz) {
    const int y0 = blockIdx.y * RB, y1 = min(y0 + RB, H);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int j = -SH2; j <= SH2; j++) {
        const int yy = min(max(y0 + j, 0), H - 1);
        const uint4 v = *reinterpret_cast<const uint4 *>(hs + (int64_t)yy * rowsz + e);
        acc.x = pk_add(acc.x, v.x); acc.y = pk_add(acc.y, v.y); acc.z = pk_add(acc.z, v.z); acc.w = pk_add(acc.w, v.w);
    }
    *reinterpret_cast<uint4 *>(C + (int64_t)y0 * rowsz + e) = acc;
    {   // headroom record: the first row of a band as upstream reaches it, C(y0) + hsum(y0-SH2-1)
        uint4 b = make_uint4(0, 0, 0, 0);
        if (y0 > 0) b = *reinterpret_cast<const uint4 *>(hs + (int64_t)max(y0 - SH2 - 1, 0) * rowsz + e);
        hmax = pk_max_u(pk_max_u(pk_add(acc.x, b.x), pk_add(acc.y, b.y)), pk_max_u(pk_add(acc.z, b.z), pk_add(acc.w, b.w)));
    }
    for (int y = y0 + 1; y < y1; y++) {
        const uint4 a = *reinterpret_cast<const uint4 *>(hs + (int64_t)min(y + SH2, H - 1) * rowsz + e);
        const uint4 b = *reinterpret_cast<const uint4 *>(hs + (int64_t)max(y - SH2 - 1, 0) * rowsz + e);
        const uint4 t = make_uint4(pk_add(acc.x, a.x), pk_add(acc.y, a.y), pk_add(acc.z, a.z), pk_add(acc.w, a.w));
        hmax = pk_max_u(hmax, pk_max_u(pk_max_u(t.x, t.y), pk_max_u(t.z, t.w)));
        acc.x = pk_sub(t.x, b.x); acc.y = pk_sub(t.y, b.y);
        acc.z = pk_sub(t.z, b.z); acc.w = pk_sub(t.w, b.w);
        *reinterpret_cast<uint4 *>(C + (int64_t)y * rowsz + e) = acc;
    }
    }
    headroom_commit_pk(hr, 0, hmax);
}

// Same vertical box sum, each hsum row read ONCE: the last 2*SH2+2 rows of the column live in a
// register ring (statically indexed: bands start at multiples of RS and the row loop is unrolled
// by RS).  Instantiated for the common block sizes; other sizes use k_vsum.
template <int SH2_, int NW /* dwords per thread: 2 or 4 */>
__global__ __launch_bounds__(256) void k_vsum_ring(const int16_t *__restrict__ hs, int16_t *__restrict__ C,
                                                   int H, int64_t rowsz, int RB /* multiple of RS */, uint32_t *hr)
{
    constexpr int RS = SH2_ <= 1 ? 4 : (SH2_ <= 3 ? 8 : (SH2_ <= 7 ? 16 : 32));  // pow2 >= 2*SH2+2
    typedef Pack<NW> V;
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * (2 * NW);  // 2*NW int16 per thread
    uint32_t hmax = 0;  // headroom record (packed running maximum)
    if (e < rowsz) {
    const int y0 = (int)blockIdx.y * RB, y1 = min(y0 + RB, H);
    V ring[RS];
    V acc;
    acc.fill(0);
    auto ld = [&](int r) {
        V v;
        v.load(hs + (int64_t)min(max(r, 0), H - 1) * rowsz + e);
        return v;
    };
    auto step = [&](const V &v, int u, int y) {  // row y = yb + u enters its window's new bottom row v
        const V &o = ring[(u - SH2_ - 1) & (RS - 1)];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const uint32_t t = pk_add(acc.r[i], v.r[i]);  // C(y-1) + hsum(y+r): an int16 lane upstream
            hmax = pk_max_u(hmax, t);
            acc.r[i] = pk_sub(t, o.r[i]);
        }
        ring[(u + SH2_) & (RS - 1)] = v;
        acc.store(C + (int64_t)y * rowsz + e);
    };
    // window of row y0: rows y0-SH2 .. y0+SH2 (y0 % RS == 0, so slot(y0 + j) = j & (RS-1) is static)
#pragma unroll
    for (int j = -SH2_; j <= SH2_; j++) {
        const V v = ld(y0 + j);
        ring[j & (RS - 1)] = v;
#pragma unroll
        for (int i = 0; i < NW; i++) acc.r[i] = pk_add(acc.r[i], v.r[i]);
    }
    acc.store(C + (int64_t)y0 * rowsz + e);
    {   // the first row of a band as upstream reaches it: C(y0-1) + hsum(y0+r) = C(y0) + hsum(y0-r-1)
        V b;
        b.fill(0);
        if (y0 > 0) b = ld(y0 - SH2_ - 1);
#pragma unroll
        for (int i = 0; i < NW; i++) hmax = pk_max_u(hmax, pk_add(acc.r[i], b.r[i]));
    }
    // a guarded block of RS rows (band edges, rows whose incoming tap is clamped)
    auto block_slow = [&](int yb) {
#pragma unroll
        for (int u = 0; u < RS; u++) {
            const int y = yb + u;
            if (y > y0 && y < y1) step(ld(y + SH2_), u, y);
        }
    };
    // fast blocks: all RS rows inside the band, no clamped tap.  Their loads are issued a whole
    // block ahead and the loop body is straight-line, so the loads stay in flight across the
    // arithmetic and stores of the previous block (a branch between issue and use would make
    // hipcc wait for vmcnt(0)).
    auto fast = [&](int yb) { return yb > y0 && yb + RS <= y1 && yb + RS - 1 + SH2_ <= H - 1; };
    auto block_load = [&](V *v, int yb) {
#pragma unroll
        for (int u = 0; u < RS; u++) v[u].load(hs + (int64_t)(yb + u + SH2_) * rowsz + e);
    };
    auto block_fast = [&](const V *v, int yb) {
#pragma unroll
        for (int u = 0; u < RS; u++) step(v[u], u, yb + u);
    };
    int yb = y0;
    block_slow(yb);
    yb += RS;
    if (fast(yb)) {
        V vA[RS], vB[RS];
        block_load(vA, yb);
        for (; fast(yb + RS) && fast(yb + 2 * RS); yb += 2 * RS) {
            block_load(vB, yb + RS);
            block_fast(vA, yb);
            block_load(vA, yb + 2 * RS);
            block_fast(vB, yb + RS);
        }
        block_fast(vA, yb);
        yb += RS;
    }
    for (; yb < y1; yb += RS) block_slow(yb);
    }
    headroom_commit_pk(hr, 0, hmax);
}

}  // namespace sgm

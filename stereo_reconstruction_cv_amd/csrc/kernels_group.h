// kernels_group.h -- small disparity ranges (D <= 64): several pixels per wavefront.
//
// With D <= 2*GW (GW = 8, 16 or 32 lanes) one pixel's disparities fill only GW lanes, so the
// path kernels of kernels_path.h / kernels_sweep.h would idle 50-87 % of every wave (the notebook's
// own setting, numDisparities = 16, main.ipynb:780, uses 8 lanes).  Here a wave is 64/GW lane
// groups, each working on an independent line (row / band) with the same instruction stream:
// the d+-1 exchange is the same DPP wave shift with the sentinel forced at group edges, minima
// are DPP butterflies that stop at the group width, addresses are per-lane buffer offsets into
// the whole volume (volumes with D <= 64 stay far below the 4 GiB a 32-bit offset reaches).
// Arithmetic and results are those of the ungrouped kernels (same path_elem / wta_pixels).
#pragma once
#include "kernels_path.h"

namespace sgm {

// whole [H][W1][D] int16 volume as one buffer resource (grouped kernels: D <= 64, < 4 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vol_rsrc(const void *p, int64_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000);
}
constexpr int SGM_OOB = 0x7ffffff0;  // per-lane offset past every volume: loads give 0, stores are dropped

// Horizontal direction (the in-row path of a pass that is not fused into a sweep: MODE_SGBM's
// fifth path with the winner-take-all), G = 64/GW rows per wave.  MODE as in k_path.
template <int GW, int MODE, bool POSW>
__global__ __launch_bounds__(64) void k_rows_g(Geom g, int rx, const int16_t *__restrict__ C, int16_t *__restrict__ S,
                                               int keepS, uint2 *__restrict__ wta)
{
    constexpr int G = 64 / GW, NP = 1, PB = 8;
    const int lane = threadIdx.x, gi = lane / GW, li = lane % GW;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int y = blockIdx.x * G + gi;
    const bool active = 2 * li < D && y < H;
    GroupEdge ge;
    ge.first = li == 0;
    ge.last = li == GW - 1;
    const int64_t vol = (int64_t)H * W1 * D * 2;
    const __amdgpu_buffer_rsrc_t Cv = vol_rsrc(C, vol), Sv = vol_rsrc(S, vol);
    const __amdgpu_buffer_rsrc_t Sst = vol_rsrc(S, (MODE != PATH_LAST || keepS) ? vol : 0);
    const int voff = active ? y * (W1 * D * 2) + li * 4 : SGM_OOB;
    const int pxb = D * 2;
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const uint32_t init = active ? 0u : SGM_SENT;
    constexpr bool READS_S = (MODE == PATH_ACCUM || MODE == PATH_LAST);
    uint2 *const wrow = wta + ((int64_t)min(y, H - 1) * g.W + g.minX1);  // per lane; idle groups never store
    const int lw = y < H ? li : GW;  // "lane" for the WTA: a group past the last row must not own a record

    Pack<NP> L;
    L.fill(init);
    ShiftRegs sr;
    Pack<NP> cA[PB], cB[PB], sA[PB], sB[PB];
    const int x0 = rx > 0 ? 0 : W1 - 1;
    auto load_t = [&](auto full_c, Pack<NP> *cb, Pack<NP> *sb, int k0) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int u = 0; u < PB; u++)
            if (FULL || k0 + u < W1) {
                const int so = (x0 + (k0 + u) * rx) * pxb;
                buf_load<NP>(cb[u], Cv, voff, so);
                if (READS_S) buf_load<NP>(sb[u], Sv, voff, so);
            }
    };
    auto pixel = [&](const Pack<NP> &cv, const Pack<NP> &sv, int k) {
        Pack<NP> Ln, Lnorm;
        uint32_t rmin;
        path_elem<NP, true, GW>(cv, L, P1s, P2s, active, Ln, rmin, sr, ge);
        const uint32_t m = group_min_pk<GW>(rmin);
        path_normalise<NP, true>(Ln, min(m & 0xffffu, m >> 16), active, Lnorm);
        Pack<NP> Sn;
        Sn.r[0] = MODE == PATH_FIRST ? Ln.r[0] : pk_adds_s(sv.r[0], Ln.r[0]);
        const int x = x0 + k * rx;
        buf_store<NP>(Sn, Sst, voff, x * pxb);
        L = Lnorm;
        return Sn;
    };
    auto compute_t = [&](auto full_c, Pack<NP> *cb, Pack<NP> *sb, int k0) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int u0 = 0; u0 < PB; u0 += 2) {
            if (FULL || k0 + u0 + 1 < W1) {
                Pack<NP> Sn[2];
                Sn[0] = pixel(cb[u0], sb[u0], k0 + u0);
                Sn[1] = pixel(cb[u0 + 1], sb[u0 + 1], k0 + u0 + 1);
                if (MODE == PATH_LAST) {
                    uint2 *recs[2] = {wrow + (x0 + (k0 + u0) * rx), wrow + (x0 + (k0 + u0 + 1) * rx)};
                    wta_pixels<NP, true, POSW, 2, GW>(Sn, lw, active, D, g.uniq, recs);
                }
            } else if (k0 + u0 < W1) {
                Pack<NP> Sn[1];
                Sn[0] = pixel(cb[u0], sb[u0], k0 + u0);
                if (MODE == PATH_LAST) {
                    uint2 *recs[1] = {wrow + (x0 + (k0 + u0) * rx)};
                    wta_pixels<NP, true, POSW, 1, GW>(Sn, lw, active, D, g.uniq, recs);
                }
            }
        }
    };
    const std::true_type full{};
    const std::false_type part{};
    int k0 = 0;
    load_t(part, cA, sA, 0);
    for (; k0 + 3 * PB <= W1; k0 += 2 * PB) {  // straight-line steady state: loads stay in flight
        load_t(full, cB, sB, k0 + PB);
        compute_t(full, cA, sA, k0);
        load_t(full, cA, sA, k0 + 2 * PB);
        compute_t(full, cB, sB, k0 + PB);
    }
    for (; k0 < W1; k0 += 2 * PB) {
        load_t(part, cB, sB, k0 + PB);
        compute_t(part, cA, sA, k0);
        load_t(part, cA, sA, k0 + 2 * PB);
        compute_t(part, cB, sB, k0 + PB);
    }
}

}  // namespace sgm

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
// GW = 64 (one row per wave, NP = 1, 2 or 4 registers per lane, PARTIAL as in k_path) is the same
// loop for large D: one buffer resource per row built once, a constant per-lane offset, a scalar
// pixel offset (k_path's general line cursor spends more scalar than vector instructions per pixel).
template <int GW, int NP, bool PARTIAL, int MODE, bool POSW>
__global__ __launch_bounds__(64) void k_rows_g(Geom g, int rx, const int16_t *__restrict__ C, int16_t *__restrict__ S,
                                               int keepS, uint2 *__restrict__ wta)
{
    // Prefetch block: the loads of block b + 1 are issued PB pixels before their first use.  These kernels run one
    // wave per SIMD (a frame has fewer rows than the GPU has SIMDs), so nothing but the prefetch distance hides memory
    // latency: with PB = 8 the D <= 128 kernels ran at exactly "latency / 8" per pixel (4K D=16: 230 ns per pixel = 1.8 us
    // of latency under load, against about 60 ns of instructions) -- round 3.  D > 128 (NP >= 2) is bound by its
    // instruction stream (WTA fused: about 250 instructions per pixel) and keeps 8.
    constexpr int G = 64 / GW, PB = NP == 1 ? 32 : 8;
    static_assert(GW == 64 || (NP == 1 && PARTIAL), "lane groups hold D <= 64: one packed register per lane");
    const int lane = threadIdx.x, gi = GW == 64 ? 0 : lane / GW, li = lane % GW;  // (GW = 64: the row must be provably uniform)
    const int W1 = g.W1, D = g.D, H = g.H;
    const int y = blockIdx.x * G + gi;
    const bool active = GW == 64 ? (!PARTIAL || 2 * NP * li < D) : (2 * NP * li < D && y < H);
    GroupEdge ge;
    ge.first = li == 0;
    ge.last = li == GW - 1;
    const int row_bytes = W1 * D * 2;
    // GW < 64: the whole volume behind one descriptor, the row in the per-lane offset (volumes of
    // D <= 64 stay below 2 GiB); GW = 64: this wave's row behind the descriptor
    const int64_t span = GW == 64 ? (int64_t)row_bytes : (int64_t)H * row_bytes;
    const int64_t rowoff = GW == 64 ? (int64_t)y * row_bytes : 0;
    const __amdgpu_buffer_rsrc_t Cv = uniform_rsrc(C, rowoff, (int)span), Sv = uniform_rsrc(S, rowoff, (int)span);
    const __amdgpu_buffer_rsrc_t Sst = Sv;
    const bool stores_S = MODE != PATH_LAST || keepS;
    const int voff = active ? (GW == 64 ? 0 : y * row_bytes) + li * NP * 4 : SGM_OOB;
    const int pxb = D * 2;
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const uint32_t init = active ? 0u : SGM_SENT;
    constexpr bool READS_S = (MODE == PATH_ACCUM || MODE == PATH_LAST);
    uint2 *const wrow = wta + ((int64_t)min(y, H - 1) * g.W + g.minX1);  // per lane; idle groups never store
    const int lw = y < H ? li : GW;  // "lane" for the WTA: a group past the last row must not own a record

    // un-normalised state U = L_r(q, .) and its minimum (both halves of ms): the form with the short
    // dependency chain between pixels (path_inner_min / path_finish, kernels_path.h)
    Pack<NP> U;
    U.fill(init);
    uint32_t ms = 0;
    ShiftRegs sr;
    uint32_t hm = 0;  // headroom record: largest min_d L_r(p, d) of this lane's row
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
        Pack<NP> t, Un;
        uint32_t rmin;
        path_inner_min<NP, PARTIAL, GW>(U, P1s, t, sr, ge);
        path_finish<NP, PARTIAL>(cv, t, ms, P2s, active, Un, rmin);
        ms = group_min_splat<GW>(rmin);
        hm = max(hm, ms);  // (splats: the unsigned maximum of splats is the splat of the maximum)
        if (PARTIAL && !active) ms = 0;  // (idle lanes: keep their sentinel arithmetic away from wrap-around)
        Pack<NP> Sn;
#pragma unroll
        for (int i = 0; i < NP; i++) Sn.r[i] = MODE == PATH_FIRST ? Un.r[i] : pk_adds_s(sv.r[i], Un.r[i]);
        const int x = x0 + k * rx;
        if (stores_S) buf_store<NP>(Sn, Sst, voff, x * pxb);
        U = Un;
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
                    wta_pixels<NP, PARTIAL, POSW, 2, GW>(Sn, lw, active, D, g.uniq, recs);
                }
            } else if (k0 + u0 < W1) {
                Pack<NP> Sn[1];
                Sn[0] = pixel(cb[u0], sb[u0], k0 + u0);
                if (MODE == PATH_LAST) {
                    uint2 *recs[1] = {wrow + (x0 + (k0 + u0) * rx)};
                    wta_pixels<NP, PARTIAL, POSW, 1, GW>(Sn, lw, active, D, g.uniq, recs);
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
    headroom_commit_pk(g.hr, 1, active ? hm : 0u);
}

// Small-D schedule (D <= 64), NO hand-off between rows: with band height 1 the boundary pre-pass leaves the
// normalised state of every row's three vertical predecessors in HBM, so a row needs nothing from its
// neighbours at run time -- no LDS ring, no lockstep, no barrier.  For small D that trade is cheap (the
// state of a row is 3*W1*D*2 bytes) and removes what bounds k_sweep there: its instruction stream per
// pixel does not shrink with D and three waves share the busiest SIMD.  The three directions from the
// previous row are then element-wise (k_vert3_g), only the in-row path is a recurrence (k_rows_g).
// (Round 1 walked all four directions along the rows in one kernel, k_rows4_g; round 2 measured the split
// form ahead -- 4K D=16: 1.25 + 0.89 -> 0.39 + 0.56 ms -- and round 3 removed the old kernel.)
// Per-row state of the small-D schedule (band height 1): one record of 3 * W1 * D int16 per row.  The grouped
// pre-pass writes it ROLE-MAJOR, [row][role][x][D]: a wave (one role, 64 / GW adjacent columns) then stores
// 64 / GW * D * 2 contiguous bytes -- whole lines; in the band layout [row][x][role][D] of the large-D
// kernels each pixel's piece is D * 2 bytes (32 at D = 16) at a stride of three: partial-line writes that
// every line of the record sees three times.  role_major = 0: the band layout, for records written by the
// single-direction kernel (debug 16).
__device__ __forceinline__ int bnd_px_off(int role_major, int W1, int x, int role)
{
    return role_major ? role * W1 + x : x * 3 + role;
}

// Boundary pre-pass of the small-D schedule (band height 1: the state after EVERY row is stored), 64/GW path
// lines per wave.  Same walk as k_prepass3 -- a wave follows its lines down the image, three roles
// each (rx = +xdir, 0, -xdir), diagonals wrap around the side border with a state reset -- but every
// lane group has its own line, so columns, wraps and addresses are per-lane values handled without
// branches; C rows are prefetched PF rows ahead through a statically indexed register ring.
// One role per wave (role = blockIdx.y, grid.y = 3): frames of small D have so few lines per SIMD that a wave's
// instruction latency bounds them (round 2 measured the three roles fused in one wave: 4K D=16 0.67-0.88 against
// 0.39 ms; that form is gone).
template <int GW>
__global__ __launch_bounds__(64) void k_prepass3_g(Geom g, int xdir, int ydir, const int16_t *__restrict__ C,
                                                   int16_t *__restrict__ bnd)
{
    constexpr int G = 64 / GW, NP = 1, PF = 24;  // rows of prefetch distance (one wave per SIMD: see k_rows_g)
    constexpr int NR = 1;  // roles handled by this wave
    constexpr int OOB = (int)0xfffffff0u;
    const int lane = threadIdx.x, gi = lane / GW, li = lane % GW;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int line = blockIdx.x * G + gi;
    const bool active = 2 * li < D && line < W1;
    GroupEdge ge;
    ge.first = li == 0;
    ge.last = li == GW - 1;
    const int pxb = D * 2, row_bytes = W1 * pxb;
    const uint32_t vol = (uint32_t)H * (uint32_t)row_bytes;
    const __amdgpu_buffer_rsrc_t Cv = __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, (int)vol, 0x00020000);
    const __amdgpu_buffer_rsrc_t Bv = __builtin_amdgcn_make_buffer_rsrc((void *)bnd, 0, (int)(3u * vol), 0x00020000);
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const uint32_t init = active ? 0u : SGM_SENT;
    const int role0 = (int)blockIdx.y;  // the role of this wave
    int rx[NR];
#pragma unroll
    for (int d = 0; d < NR; d++) rx[d] = (role0 + d) == 0 ? xdir : ((role0 + d) == 1 ? 0 : -xdir);
    const int y0 = ydir > 0 ? 0 : H - 1;
    int xl[NR], xc[NR];  // per lane: column of the load cursor (PF rows ahead) and of the compute cursor
#pragma unroll
    for (int d = 0; d < NR; d++) xl[d] = xc[d] = min(line, W1 - 1);
    auto wrapped = [&](int &x) {  // left the image -> re-enter on the other side
        const bool w = x >= W1 || x < 0;
        x = x >= W1 ? 0 : (x < 0 ? W1 - 1 : x);
        return w;
    };
    Pack<NP> L[NR], cv[PF][NR];
#pragma unroll
    for (int d = 0; d < NR; d++) L[d].fill(init);
    ShiftRegs sr[NR];
    uint32_t hm = 0;
    auto issue = [&](Pack<NP> *c3, int s) {  // C of sweep-order row s at the load cursors, then advance them
        const int rowoff = (y0 + s * ydir) * row_bytes;
#pragma unroll
        for (int d = 0; d < NR; d++) {
            const int off = (active && s < H) ? rowoff + xl[d] * pxb + li * 4 : OOB;
            buf_load<NP>(c3[d], Cv, off, 0);
            xl[d] += rx[d];
            wrapped(xl[d]);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; u++) issue(cv[u], u);
    auto row = [&](int u, int s) {
        Pack<NP> N[NR];
        uint32_t r[NR];
#pragma unroll
        for (int d = 0; d < NR; d++) path_elem<NP, true, GW>(cv[u][d], L[d], P1s, P2s, active, N[d], r[d], sr[d], ge);
        const uint32_t m0s = group_min_splat<GW>(r[0]);
        if (s < H) hm = max(hm, m0s & 0xffffu);
        path_normalise_splat<NP, true>(N[0], m0s, active, L[0]);
        // the state the row s + 1 will read: bnd[s + 1][role][column][D] (role-major: bnd_px_off)
#pragma unroll
        for (int d = 0; d < NR; d++) {
            const int off = (active && s + 1 < H) ? (int)(((uint32_t)(s + 1) * 3u + (uint32_t)(role0 + d)) * (uint32_t)W1 + (uint32_t)xc[d]) * pxb + li * 4 : OOB;
            buf_store<NP>(L[d], Bv, off, 0);
            xc[d] += rx[d];
            if (wrapped(xc[d])) L[d].fill(init);  // (per lane: a select)
        }
        issue(cv[u], s + PF);
    };
    int s0 = 0;
    for (; s0 + PF <= H; s0 += PF) {  // straight-line: the loads issued here are used PF rows later
#pragma unroll
        for (int u = 0; u < PF; u++) row(u, s0 + u);
    }
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (s0 + u < H) row(u, s0 + u);
    headroom_commit_pk(g.hr, 1, active ? hm : 0u);
}

// The three directions that come from the previous row, for EVERY pixel at once (D <= 64, band height 1):
// with the boundary pre-pass having left every row's three predecessor states in HBM, these paths need
// no recurrence here at all -- N_r(p, d) = C(p, d) + min(Q_r(d), Q_r(d +- 1) + P1, P2) is element-wise --
// so the kernel is a plain streaming pass over all pixels (one lane group per pixel) instead of a walk
// along the rows: S = sat(N_A + N_B + N_C [+ S]).  The in-row direction then runs as k_rows_g (ACCUM).
// MODE: PATH_FIRST / PATH_ACCUM.
template <int GW, int MODE>
__global__ __launch_bounds__(256) void k_vert3_g(Geom g, int xdir, int ydir, const int16_t *__restrict__ C,
                                                 int16_t *__restrict__ S, const int16_t *__restrict__ bnd,
                                                 int role_major /* layout of bnd: see bnd_px_off */)
{
    constexpr int G = 64 / GW, NP = 1;
    constexpr int OOB = (int)0xfffffff0u;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gi = lane / GW, li = lane % GW;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int j = blockIdx.y;  // row index in sweep order
    const int y = ydir > 0 ? j : H - 1 - j;
    const int xb = (blockIdx.x * 4 + wv) * 64;  // this wave's 64 consecutive columns
    GroupEdge ge;
    ge.first = li == 0;
    ge.last = li == GW - 1;
    const int pxb = D * 2, row_bytes = W1 * pxb;
    const uint32_t vol = (uint32_t)H * (uint32_t)row_bytes;
    const __amdgpu_buffer_rsrc_t Cv = __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, (int)vol, 0x00020000);
    const __amdgpu_buffer_rsrc_t Sv = __builtin_amdgcn_make_buffer_rsrc((void *)S, 0, (int)vol, 0x00020000);
    const __amdgpu_buffer_rsrc_t Bv = __builtin_amdgcn_make_buffer_rsrc((void *)bnd, 0, (int)(3u * vol), 0x00020000);
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const bool lanes = 2 * li < D;
    const bool has_prev = j > 0;
    const uint32_t rowoff = (uint32_t)y * (uint32_t)row_bytes, brow = (uint32_t)j * 3u * (uint32_t)row_bytes;
    ShiftRegs srA, srB, srC;
#pragma unroll 4
    for (int it = 0; it < GW; it++) {  // G pixels per iteration, 64 per wave
        const int x = xb + it * G + gi;
        const bool active = lanes && x < W1;
        const int k = xdir > 0 ? x : W1 - 1 - x;  // position in the sweep's order: role A comes from k - 1, C from k + 1
        const int xa = min(max(x - xdir, 0), W1 - 1), xc = min(max(x + xdir, 0), W1 - 1);
        const int voff = active ? (int)(rowoff + (uint32_t)x * pxb) + li * 4 : OOB;
        const bool rd = active && has_prev;
        Pack<NP> c, sp, QA, QB, QC;
        buf_load<NP>(c, Cv, voff, 0);
        if (MODE == PATH_ACCUM) buf_load<NP>(sp, Sv, voff, 0);
        buf_load<NP>(QA, Bv, rd ? (int)(brow + (uint32_t)bnd_px_off(role_major, W1, xa, 0) * pxb) + li * 4 : OOB, 0);
        buf_load<NP>(QB, Bv, rd ? (int)(brow + (uint32_t)bnd_px_off(role_major, W1, x, 1) * pxb) + li * 4 : OOB, 0);
        buf_load<NP>(QC, Bv, rd ? (int)(brow + (uint32_t)bnd_px_off(role_major, W1, xc, 2) * pxb) + li * 4 : OOB, 0);
        // out-of-image predecessors and idle lanes: start state / sentinel
        if (k == 0) QA.fill(0u);
        if (k == W1 - 1) QC.fill(0u);
        if (!active) {
            QA.fill(SGM_SENT);
            QB.fill(SGM_SENT);
            QC.fill(SGM_SENT);
        }
        Pack<NP> NA, NB, NC;
        uint32_t rA, rB, rC;
        path_elem<NP, true, GW>(c, QA, P1s, P2s, active, NA, rA, srA, ge);
        path_elem<NP, true, GW>(c, QB, P1s, P2s, active, NB, rB, srB, ge);
        path_elem<NP, true, GW>(c, QC, P1s, P2s, active, NC, rC, srC, ge);
        Pack<NP> Sn;
        uint32_t v = pk_adds_s(pk_adds_s(NA.r[0], NB.r[0]), NC.r[0]);
        if (MODE == PATH_ACCUM) v = pk_adds_s(v, sp.r[0]);
        Sn.r[0] = v;
        buf_store<NP>(Sn, Sv, voff, 0);
    }
}

}  // namespace sgm

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
// `unit`: which group of G rows this wave takes (k_rows_g: blockIdx.x; k_paths5_g: its share of that launch).
template <int GW, int NP, bool PARTIAL, int MODE, bool POSW>
__device__ __forceinline__ void rows_g_body(const Geom &g, int rx, const int16_t *__restrict__ C, int16_t *__restrict__ S,
                                            int keepS, uint2 *__restrict__ wta, int unit)
{
    // Prefetch block: the loads of block b + 1 are issued PB pixels before their first use.  These kernels run one
    // wave per SIMD (a frame has fewer rows than the GPU has SIMDs), so nothing but the prefetch distance hides memory
    // latency: with PB = 8 the D <= 128 kernels ran at exactly "latency / 8" per pixel (4K D=16: 230 ns per pixel = 1.8 us
    // of latency under load, against about 60 ns of instructions) -- round 3.  D > 128 (NP >= 2) is bound by its
    // instruction stream (WTA fused: about 250 instructions per pixel) and keeps 8.
    constexpr int G = 64 / GW, PB = NP == 1 ? 32 : 8;
    constexpr bool WTA_PART = PARTIAL || GW < 64;  // (the winner-take-all keeps its masks: a group past the last row must store nothing)
    // GW < 64: PARTIAL says D < 2 * GW (D = 48 in groups of 32 lanes; 16, 32 and 64 fill their groups).  Full groups need
    // no sentinel selects: a group without a row (past H) works on the zeros its out-of-range loads return, its stores
    // are dropped, it owns no winner-take-all record (lw) and its headroom is masked at the end -- two v_cndmask
    // per step less in a kernel that is bound by its own instruction stream.
    static_assert(GW == 64 || NP == 1, "lane groups hold D <= 64: one packed register per lane");
    const int lane = threadIdx.x & 63, gi = GW == 64 ? 0 : lane / GW, li = lane % GW;  // (GW = 64: the row must be provably uniform)
    const int W1 = g.W1, D = g.D, H = g.H;
    const int y = unit * G + gi;
    const bool active = GW == 64 ? (!PARTIAL || 2 * NP * li < D) : (2 * NP * li < D && y < H);
    GroupEdge ge;
    ge.set(li, GW);
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
    // Byte offsets of a block's pixels.  A single wave issues one instruction of ANY kind every four cycles, so the
    // scalar arithmetic of "(x0 + (k0 + u) * rx) * pxb" per pixel (or, hoisted out of the loop by the compiler, 2 * PB
    // products that no longer fit the SGPRs and come back through v_readlane) costs these kernels as much as vector work.
    //  * full lane groups (IMM: D = 2 GW, a pixel is 4 GW bytes): the block's base is one scalar, the pixel is the
    //    instruction's immediate offset (PB * 4 GW <= 4096); leftwards the base is the block's last pixel.  The two
    //    directions are two copies of the steady-state loop under one uniform branch;
    //  * otherwise (and in the partial blocks at the row end): base + u * step with the step made opaque once per block.
    constexpr bool IMM = GW < 64 && !PARTIAL;
    constexpr int PXB = 4 * GW;
    static_assert(!IMM || (PB - 1) * PXB + 4 <= 4096, "immediate offsets of a block");
    const int so_step0 = rx * pxb;
    auto block_step = [&]() __attribute__((always_inline)) {
        int s = so_step0;
        if constexpr (!IMM) asm volatile("" : "+s"(s));  // (IMM: only the partial blocks at the row end come here)
        return s;
    };
    // FWD: compile-time direction of the IMM form (+1: true); GEN: the general form
    auto load_t = [&](auto full_c, auto fwd_c, Pack<NP> *cb, Pack<NP> *sb, int k0) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value, FWD = decltype(fwd_c)::value;
        if constexpr (IMM && FULL) {
            const int sb0 = (x0 + (FWD ? k0 : -(k0 + PB - 1))) * PXB;  // lowest address of the block
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int imm = (FWD ? u : PB - 1 - u) * PXB;
                buf_load<NP>(cb[u], Cv, voff + imm, sb0);
                if (READS_S) buf_load<NP>(sb[u], Sv, voff + imm, sb0);
            }
        } else {
            const int step = block_step(), so0 = (x0 + k0 * rx) * pxb;
#pragma unroll
            for (int u = 0; u < PB; u++)
                if (FULL || k0 + u < W1) {
                    const int so = so0 + u * step;
                    buf_load<NP>(cb[u], Cv, voff, so);
                    if (READS_S) buf_load<NP>(sb[u], Sv, voff, so);
                }
        }
    };
    auto pixel = [&](const Pack<NP> &cv, const Pack<NP> &sv, int vo, int so) __attribute__((always_inline)) {
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
        if (stores_S) buf_store<NP>(Sn, Sst, vo, so);
        U = Un;
        return Sn;
    };
    auto compute_t = [&](auto full_c, auto fwd_c, Pack<NP> *cb, Pack<NP> *sb, int k0) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value, FWD = decltype(fwd_c)::value;
        constexpr bool BI = IMM && FULL;
        const int step = BI ? 0 : block_step();
        const int so0 = BI ? (x0 + (FWD ? k0 : -(k0 + PB - 1))) * PXB : (x0 + k0 * rx) * pxb;
        auto vo = [&](int u) __attribute__((always_inline)) { return BI ? voff + (FWD ? u : PB - 1 - u) * PXB : voff; };
        auto so = [&](int u) __attribute__((always_inline)) { return BI ? so0 : so0 + u * step; };
#pragma unroll
        for (int u0 = 0; u0 < PB; u0 += 2) {
            if (FULL || k0 + u0 + 1 < W1) {
                Pack<NP> Sn[2];
                Sn[0] = pixel(cb[u0], sb[u0], vo(u0), so(u0));
                Sn[1] = pixel(cb[u0 + 1], sb[u0 + 1], vo(u0 + 1), so(u0 + 1));
                if (MODE == PATH_LAST) {
                    uint2 *recs[2] = {wrow + (x0 + (k0 + u0) * rx), wrow + (x0 + (k0 + u0 + 1) * rx)};
                    wta_pixels<NP, WTA_PART, POSW, 2, GW>(Sn, lw, active, D, g.uniq, recs);
                }
            } else if (k0 + u0 < W1) {
                Pack<NP> Sn[1];
                Sn[0] = pixel(cb[u0], sb[u0], vo(u0), so(u0));
                if (MODE == PATH_LAST) {
                    uint2 *recs[1] = {wrow + (x0 + (k0 + u0) * rx)};
                    wta_pixels<NP, WTA_PART, POSW, 1, GW>(Sn, lw, active, D, g.uniq, recs);
                }
            }
        }
    };
    const std::true_type full{};
    const std::false_type part{};
    int k0 = 0;
    load_t(part, part, cA, sA, 0);
    auto steady = [&](auto fwd_c) __attribute__((always_inline)) {
        for (; k0 + 3 * PB <= W1; k0 += 2 * PB) {  // straight-line steady state: loads stay in flight
            load_t(full, fwd_c, cB, sB, k0 + PB);
            compute_t(full, fwd_c, cA, sA, k0);
            load_t(full, fwd_c, cA, sA, k0 + 2 * PB);
            compute_t(full, fwd_c, cB, sB, k0 + PB);
        }
    };
    if (!IMM || rx > 0) steady(full);
    else steady(part);
    for (; k0 < W1; k0 += 2 * PB) {
        load_t(part, part, cB, sB, k0 + PB);
        compute_t(part, part, cA, sA, k0);
        load_t(part, part, cA, sA, k0 + 2 * PB);
        compute_t(part, part, cB, sB, k0 + PB);
    }
    headroom_commit_pk(g.hr, 1, active ? hm : 0u);
}
template <int GW, int NP, bool PARTIAL, int MODE, bool POSW>
__global__ __launch_bounds__(64) void k_rows_g(Geom g, int rx, const int16_t *__restrict__ C, int16_t *__restrict__ S,
                                               int keepS, uint2 *__restrict__ wta)
{
    rows_g_body<GW, NP, PARTIAL, MODE, POSW>(g, rx, C, S, keepS, wta, (int)blockIdx.x);
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
// lines per wave.  Same walk as k_prepass3 -- a wave follows its lines down the image, one role each (rx = +xdir, 0,
// -xdir; role = blockIdx.y: frames of small D have so few lines per SIMD that a wave's instruction latency bounds
// them, round 2 measured the three roles fused in one wave at 0.67-0.88 against 0.39 ms for 4K D=16), diagonals wrap
// around the side border with a state reset -- but every lane group has its own line.
//
// This kernel runs about one wave per SIMD, each at the issue rate of its own instruction stream, so the step is
// written for its instruction count (round 3: 39 -> 21 vector instructions per step at GW = 8):
//  * cursors are per-lane BYTE OFFSETS advanced by a per-lane constant (one v_add each for the load cursor, PF rows
//    ahead, and the store cursor); idle lanes hold an out-of-range offset and a step of 0;
//  * rows past the end of the image need no test: C sits behind a descriptor of exactly H rows and the record behind
//    one of exactly H rows of state, so the prefetch past the last row reads zeros and the state "after the last row"
//    is dropped by the bounds check;
//  * a line's wrap around the side border is found on the SCALAR unit: the lines of a wave are adjacent, so they
//    wrap in G consecutive steps out of W1 -- a counter modulo W1 per cursor says when (q < G), and only then a
//    per-lane compare picks the lane group that wraps (offset -+ one row, state reset);
//  * PARTIAL (D < 2 GW, i.e. D = 48) keeps the sentinel selects of idle lanes; full groups have none (k_rows_g).
// C rows are prefetched PF rows ahead through a statically indexed register ring.
//
// VOLS = false: the pre-pass proper -- the NORMALISED state after row s goes to the record of row s + 1 (role-major,
// bnd_px_off), k_vert3_g reads it.  VOLS = true (k_paths5_g, MODE_SGBM): the walk IS the aggregation of its direction --
// the un-normalised L_r(p, .) it forms on the way is exactly what upstream adds to S, so each role writes it to a volume
// of its own ([y][x][d], `out`) and the winner-take-all adds the volumes: no record, no element-wise kernel (round 3:
// 3 V written + 3 V read + 1 V of C read again + the kernel itself off the critical path of a small frame).
template <int GW, bool PARTIAL, bool VOLS>
__device__ __forceinline__ void lines3_g_body(const Geom &g, int xdir, int ydir, const int16_t *__restrict__ C,
                                              int16_t *__restrict__ out, int unit, int role)
{
    constexpr int G = 64 / GW, NP = 1, PF = 24;  // rows of prefetch distance (one wave per SIMD: see k_rows_g)
    constexpr uint32_t OOB = 0xfffffff0u;
    const int lane = threadIdx.x & 63, gi = lane / GW, li = lane % GW;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int base = unit * G;  // first line of this wave
    const int line = base + gi;
    const bool active = (!PARTIAL || 2 * li < D) && line < W1;
    GroupEdge ge;
    ge.set(li, GW);
    const int pxb = D * 2, row_bytes = W1 * pxb;
    const uint32_t vol = (uint32_t)H * (uint32_t)row_bytes;
    const __amdgpu_buffer_rsrc_t Cv = __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, (int)vol, 0x00020000);
    const __amdgpu_buffer_rsrc_t Bv = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)(VOLS ? vol : 3u * vol), 0x00020000);
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const uint32_t init = (PARTIAL && !active) ? SGM_SENT : 0u;
    const int rx = role == 0 ? xdir : (role == 1 ? 0 : -xdir);
    const int y0 = ydir > 0 ? 0 : H - 1;
    const int xs = min(line, W1 - 1);
    // load cursor: C(row y0 + k ydir, column of the line after k steps); store cursor: record row s + 1 (sweep order), this
    // role, the column the state was computed at (role-major record: bnd_px_off)
    uint32_t offL = active ? (uint32_t)y0 * (uint32_t)row_bytes + (uint32_t)(xs * pxb + li * 4) : OOB;
    uint32_t offB = !active ? OOB : (VOLS ? offL : (uint32_t)(3 + role) * (uint32_t)row_bytes + (uint32_t)(xs * pxb + li * 4));
    const uint32_t dL = active ? (uint32_t)(ydir * row_bytes + rx * pxb) : 0u;
    const uint32_t dB = VOLS ? dL : (active ? (uint32_t)(3 * row_bytes + rx * pxb) : 0u);
    const uint32_t wadj = active ? (uint32_t)(-rx * row_bytes) : 0u;  // a wrap: the column jumps by -+ W1
    // Wrap counters (scalar, modulo W1): the lane group gi wraps at the advance where q == tgt; tgt < G for every group.
    //   rx = +1: column line + k reaches W1        <=>  (k + base + G - 1) mod W1 == (G - 1 - gi) mod W1
    //   rx = -1: column line - k reaches -1        <=>  (k - base - 1)     mod W1 == gi mod W1
    // (k = 1 for the first advance.)  qL runs with the load cursor, qC with the store cursor.
    const int tgt = rx > 0 ? (G - 1 - gi) % W1 : gi % W1;
    int qC = rx > 0 ? (base + G - 1) % W1 : ((-base - 1) % W1 + W1) % W1;  // k = 0
    int qL = qC;
    Pack<NP> L, cv[PF];
    L.fill(init);
    ShiftRegs sr;
    uint32_t hm = 0;
    auto run = [&](auto wraps_c) __attribute__((always_inline)) {  // (role 1 walks straight down: no wrap logic at all)
        constexpr bool WRAPS = decltype(wraps_c)::value;
        auto advance = [&](uint32_t &off, uint32_t d, int &q) __attribute__((always_inline)) -> bool {  // true: some lane group of this wave wraps now
            off += d;
            if (!WRAPS) return false;
            q = q + 1 == W1 ? 0 : q + 1;
            return q < G;
        };
        auto issue = [&](Pack<NP> &c) __attribute__((always_inline)) {  // C at the load cursor, then advance it
            buf_load<NP>(c, Cv, (int)offL, 0);
            if (advance(offL, dL, qL)) {
                asm volatile("");  // (a real branch: G steps out of W1 come here; if-converted, its compare and selects ran every step)
                offL += (qL == tgt) ? wadj : 0u;
            }
        };
#pragma unroll
        for (int u = 0; u < PF; u++) issue(cv[u]);
        auto row = [&](int u) __attribute__((always_inline)) {
            Pack<NP> N;
            uint32_t r;
            path_elem<NP, PARTIAL, GW>(cv[u], L, P1s, P2s, active, N, r, sr, ge);
            const uint32_t m0s = group_min_splat<GW>(r);
            hm = max(hm, m0s);  // (splats: the unsigned maximum of splats is the splat of the maximum)
            path_normalise_splat<NP, PARTIAL>(N, m0s, active, L);
            buf_store<NP>(VOLS ? N : L, Bv, (int)offB, 0);
            if (advance(offB, dB, qC)) {
                asm volatile("");
                const bool w = qC == tgt;
                offB += w ? wadj : 0u;
                if (w) L.fill(init);  // (per lane: a select)
            }
            issue(cv[u]);
        };
        int s0 = 0;
        for (; s0 + PF <= H; s0 += PF) {  // straight-line: the loads issued here are used PF rows later
#pragma unroll
            for (int u = 0; u < PF; u++) row(u);
        }
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (s0 + u < H) row(u);
    };
    if (rx == 0) run(std::false_type{});
    else run(std::true_type{});
    headroom_commit_pk(g.hr, 1, active ? hm : 0u);
}
template <int GW, bool PARTIAL>
__global__ __launch_bounds__(64) void k_prepass3_g(Geom g, int xdir, int ydir, const int16_t *__restrict__ C,
                                                   int16_t *__restrict__ bnd)
{
    lines3_g_body<GW, PARTIAL, false>(g, xdir, ydir, C, bnd, (int)blockIdx.x, (int)blockIdx.y);
}
// All five directions of MODE_SGBM for D <= 64 in ONE launch, each into a volume of its own (the winner-take-all adds
// them): workgroups [0, 2 nr) walk the rows (even: right to left -> SW, odd: left to right -> SE; nr = groups of G rows),
// the rest walk the lines of the three directions from the row above (role = index mod 3 -> SA / SB / SC).
// One launch because of where single-wave workgroups land: launches that run side by side on streams of their own are
// each dealt out from the same first CU and SIMD onwards, so the 270 waves of one in-row pass (4K, D = 16) shared
// their SIMDs with the 270 of the other while three quarters of the chip sat idle -- two in-row passes side by side took
// 0.37 ms against 0.23 ms for one alone, and 0.58 ms with the line walk beside them.  Dealt out as one grid the waves
// spread over all SIMDs; the longest chains (the rows: W1 steps) come first.
template <int GW, bool PARTIAL>
__global__ __launch_bounds__(64) void k_paths5_g(Geom g, int xdir, int ydir, const int16_t *__restrict__ C,
                                                 int16_t *__restrict__ SA, int16_t *__restrict__ SB, int16_t *__restrict__ SC,
                                                 int16_t *__restrict__ SW, int16_t *__restrict__ SE, int nr)
{
    const int b = (int)blockIdx.x;
    if (b < 2 * nr) {
        rows_g_body<GW, 1, PARTIAL, PATH_FIRST, true>(g, (b & 1) ? +1 : -1, C, (b & 1) ? SE : SW, 1, nullptr, b >> 1);
    } else {
        const int i = b - 2 * nr, role = i % 3;
        lines3_g_body<GW, PARTIAL, true>(g, xdir, ydir, C, role == 0 ? SA : (role == 1 ? SB : SC), i / 3, role);
    }
}

// The three directions that come from the previous row, for EVERY pixel at once (D <= 64, band height 1):
// with the boundary pre-pass having left every row's three predecessor states in HBM, these paths need
// no recurrence here at all -- N_r(p, d) = C(p, d) + min(Q_r(d), Q_r(d +- 1) + P1, P2) is element-wise --
// so the kernel is a plain streaming pass over all pixels (one lane group per pixel) instead of a walk
// along the rows: S = sat(N_A + N_B + N_C [+ S]).  The in-row direction then runs as k_rows_g (ACCUM).
// MODE: PATH_FIRST / PATH_ACCUM.
// Written for its instruction count like the kernels above (round 3: 74 -> about 40 vector instructions per G pixels):
// the row of C / S and each role's row of the record sit behind descriptors of their own, so a pixel past the row end
// and a predecessor outside the image (x -+ 1 = -1 or W1: the path starts there, state 0) are simply out of range --
// loads return 0, stores are dropped, no compare, no select; the three offsets advance by G pixels per iteration.
template <int GW, int MODE, bool PARTIAL>
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
    ge.set(li, GW);
    const int pxb = D * 2, row_bytes = W1 * pxb;
    const __amdgpu_buffer_rsrc_t Cv = uniform_rsrc(C, (int64_t)y * row_bytes, row_bytes);
    const __amdgpu_buffer_rsrc_t Sv = uniform_rsrc(S, (int64_t)y * row_bytes, row_bytes);
    // record of row j (the state the row above left), one descriptor per role; row 0 has no predecessors: empty range.
    // role-major [role][x][D]: role r starts r rows in, a pixel is pxb apart; band layout [x][role][D]: r pixels in, 3 pxb apart
    const int rsz = j > 0 ? (role_major ? row_bytes : 3 * row_bytes) : 0;
    const int rstep = role_major ? row_bytes : pxb, qpx = role_major ? pxb : 3 * pxb;
    const int64_t rbase = (int64_t)j * 3 * row_bytes;
    const __amdgpu_buffer_rsrc_t BA = uniform_rsrc(bnd, rbase, rsz);
    const __amdgpu_buffer_rsrc_t BB = uniform_rsrc(bnd, rbase + rstep, max(rsz - (role_major ? 0 : rstep), 0));
    const __amdgpu_buffer_rsrc_t BC = uniform_rsrc(bnd, rbase + 2 * rstep, max(rsz - (role_major ? 0 : 2 * rstep), 0));
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const bool lanes = !PARTIAL || 2 * li < D;
    // role A comes from column x - xdir, C from x + xdir (k_prepass3_g's roles)
    const int x0 = xb + gi;
    int voff = lanes ? x0 * pxb + li * 4 : OOB;
    int qa = lanes ? (x0 - xdir) * qpx + li * 4 : OOB, qb = lanes ? x0 * qpx + li * 4 : OOB, qc = lanes ? (x0 + xdir) * qpx + li * 4 : OOB;
    const int dv = lanes ? G * pxb : 0, dq = lanes ? G * qpx : 0;
    ShiftRegs srA, srB, srC;
#pragma unroll 4
    for (int it = 0; it < GW; it++) {  // G pixels per iteration, 64 per wave
        Pack<NP> c, sp, QA, QB, QC;
        buf_load<NP>(c, Cv, voff, 0);
        if (MODE == PATH_ACCUM) buf_load<NP>(sp, Sv, voff, 0);
        buf_load<NP>(QA, BA, qa, 0);
        buf_load<NP>(QB, BB, qb, 0);
        buf_load<NP>(QC, BC, qc, 0);
        if (PARTIAL && !lanes) {  // idle lanes: sentinel
            QA.fill(SGM_SENT);
            QB.fill(SGM_SENT);
            QC.fill(SGM_SENT);
        }
        Pack<NP> NA, NB, NC;
        uint32_t rA, rB, rC;
        path_elem<NP, PARTIAL, GW>(c, QA, P1s, P2s, lanes, NA, rA, srA, ge);
        path_elem<NP, PARTIAL, GW>(c, QB, P1s, P2s, lanes, NB, rB, srB, ge);
        path_elem<NP, PARTIAL, GW>(c, QC, P1s, P2s, lanes, NC, rC, srC, ge);
        Pack<NP> Sn;
        uint32_t v = pk_adds_s(pk_adds_s(NA.r[0], NB.r[0]), NC.r[0]);
        if (MODE == PATH_ACCUM) v = pk_adds_s(v, sp.r[0]);
        Sn.r[0] = v;
        buf_store<NP>(Sn, Sv, voff, 0);
        voff += dv;
        qa += dq;
        qb += dq;
        qc += dq;
    }
}

}  // namespace sgm

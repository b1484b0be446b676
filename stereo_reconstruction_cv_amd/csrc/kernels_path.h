// kernels_path.h -- semi-global path aggregation, one wavefront per path line.
//
// Replaces the per-row L_r recurrences and the winner-take-all scan of upstream
// computeDisparitySGBM (behind stereo.compute, /root/reference/main.ipynb:668; arithmetic in
// SURVEY.md A.5/A.6 and oracle/sgbm_oracle.c: path_step and the selection loop of sgbm_core).
//
//   L_r(p,d) = C(p,d) + min(L_r(q,d), L_r(q,d-1)+P1, L_r(q,d+1)+P1, min_k L_r(q,k)+P2) - min_k L_r(q,k)
//
// with q = p - r and zero state when q leaves the valid-column domain.  Every direction is an
// independent set of lines: rows for the two horizontal directions; for the six directions with
// a vertical component, W1 lines that start in the first processed row and move one row per
// step, diagonals wrapping around the side border with a state reset (each (row, column) is
// visited by exactly one line).  The upstream pass structure (4 directions fused per row) is a
// CPU scheduling choice; the sums S are order-free because every L >= 0 and S saturates.
//
// A wavefront keeps L_r(q, .) in registers (NP packed pairs per lane), gets d-1 / d+1 through
// DPP wave shifts and the minimum through a DPP butterfly + readlane; cost rows are prefetched
// a block of steps ahead into registers.
#pragma once
#include "kernels_cost.h"

namespace sgm {

enum { PATH_FIRST = 0, PATH_ACCUM = 1, PATH_LAST = 2, PATH_BOUNDARY = 3 };

// One step of the recurrence for the D disparities a wavefront holds, on NORMALISED state:
// the wave carries Lq'(d) = L_r(q,d) - min_k L_r(q,k), for which the recurrence reads
//     L_r(p,d) = C(p,d) + min(Lq'(d), Lq'(d-1)+P1, Lq'(d+1)+P1, P2)
// (no minimum needed inside the element-wise part).  Returns the un-normalised L_r(p,.), which
// is what S accumulates, and the per-lane partial minimum (both int16 halves) for the caller to
// reduce -- alone (wave_min_pk) or batched with other directions.  Idle lanes of a partial wave
// hold the MAX_COST sentinel.
// Persistent "fill" registers of the two DPP wave shifts of one direction: lane 0 of `up` and
// lane 63 of `dn` never receive data from a neighbour, so they keep the MAX_COST sentinel they
// were initialised with for the whole kernel; tying the shift's `old` operand to the previous
// value of the same variable lets hipcc emit the bare v_mov_b32_dpp without a refill.
struct ShiftRegs {
    uint32_t up = SGM_SENT, dn = SGM_SENT;
};

// Lane groups (GW < 64, several pixels per wave): first / last say whether this lane is the first /
// last of its group; those lanes must see the sentinel, not the neighbouring group's value.
// Kept as per-lane WORDS, {MAX_COST, MAX_COST} on the edge lane and 0 elsewhere: path states are packed pairs of values in
// [0, 0x7fff] (the int16 regime; wave_min4_splat), so the unsigned 32-bit maximum of a shifted-in state and the edge
// word is the sentinel on an edge lane and the state elsewhere -- and that maximum takes the DPP shift as its
// operand: ONE v_max_u32_dpp per shift instead of v_mov_b32_dpp + v_cndmask_b32.  (bound_ctrl: lanes 0 / 63,
// which have no source lane, read 0 and are edge lanes of their groups.)  The small-D kernels are bound by the
// instruction stream of a single wave: every instruction per step counts there.
struct GroupEdge {
    uint32_t first = 0, last = 0;
    __device__ __forceinline__ void set(int li, int GW)
    {
        first = li == 0 ? SGM_SENT : 0u;
        last = li == GW - 1 ? SGM_SENT : 0u;
    }
};
// the value a lane group's pixel sees one lane down / up, edges holding the sentinel
template <int GW> __device__ __forceinline__ uint32_t group_from_lower(uint32_t v, const GroupEdge &ge)
{
    return max(dpp_view<DPP_WAVE_SHR1>(v), ge.first);
}
template <int GW> __device__ __forceinline__ uint32_t group_from_upper(uint32_t v, const GroupEdge &ge)
{
    return max(dpp_view<DPP_WAVE_SHL1>(v), ge.last);
}
template <int NP, bool PARTIAL, int GW = 64>
__device__ __forceinline__ void path_elem(const Pack<NP> &Cp, const Pack<NP> &Lq, uint32_t P1s, uint32_t P2s,
                                          bool active, Pack<NP> &Ln, uint32_t &rmin, ShiftRegs &sr,
                                          GroupEdge ge = GroupEdge())
{
    uint32_t up, dn;
    if constexpr (GW < 64) {
        up = group_from_lower<GW>(Lq.r[NP - 1], ge);
        dn = group_from_upper<GW>(Lq.r[0], ge);
    } else {
        sr.up = from_lower_lane(Lq.r[NP - 1], sr.up);
        sr.dn = from_upper_lane(Lq.r[0], sr.dn);
        up = sr.up;
        dn = sr.dn;
    }
    rmin = SGM_SENT;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint32_t prevp = i == 0 ? up : Lq.r[i - 1];
        const uint32_t nextp = i == NP - 1 ? dn : Lq.r[i + 1];
        const uint32_t lm1 = __builtin_amdgcn_alignbit(Lq.r[i], prevp, 16);
        const uint32_t lp1 = __builtin_amdgcn_alignbit(nextp, Lq.r[i], 16);
        uint32_t t = pk_adds_s(pk_min_s(lm1, lp1), P1s);
        t = pk_min_s(pk_min_s(t, Lq.r[i]), P2s);
        uint32_t v = pk_add(Cp.r[i], t);
        if (PARTIAL) v = active ? v : SGM_SENT;
        Ln.r[i] = v;
        rmin = pk_min_s(rmin, v);
    }
}

// The same recurrence on UN-normalised state, for kernels whose speed is the latency of one line's
// dependency chain (the in-row kernels k_rows_g).  With U = L_r(q, .) itself and m = min_k U(k):
//     L_r(p,d) = C(p,d) + min( min(U(d), U(d-1)+P1, U(d+1)+P1) - m, P2 )
// The inner minimum t(d) does not need m, so it is computed while the reduction that yields m is still
// running; only "t - m, min P2, + C" (3 dependent operations instead of 9) sit between one pixel's
// reduction and the next.  Exact in integers: t >= m, and a saturated U(d+-1)+P1 never wins the
// minimum against U(d) <= 32767.
template <int NP, bool PARTIAL, int GW = 64>
__device__ __forceinline__ void path_inner_min(const Pack<NP> &U, uint32_t P1s, Pack<NP> &t, ShiftRegs &sr,
                                               GroupEdge ge = GroupEdge())
{
    uint32_t up, dn;
    if constexpr (GW < 64) {
        up = group_from_lower<GW>(U.r[NP - 1], ge);
        dn = group_from_upper<GW>(U.r[0], ge);
    } else {
        sr.up = from_lower_lane(U.r[NP - 1], sr.up);
        sr.dn = from_upper_lane(U.r[0], sr.dn);
        up = sr.up;
        dn = sr.dn;
    }
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint32_t prevp = i == 0 ? up : U.r[i - 1];
        const uint32_t nextp = i == NP - 1 ? dn : U.r[i + 1];
        const uint32_t lm1 = __builtin_amdgcn_alignbit(U.r[i], prevp, 16);
        const uint32_t lp1 = __builtin_amdgcn_alignbit(nextp, U.r[i], 16);
        t.r[i] = pk_min_s(pk_adds_s(pk_min_s(lm1, lp1), P1s), U.r[i]);
    }
}
// U(p, .) from t, the splat minimum ms = {m, m} of the previous pixel and the cost; rmin as in path_elem
template <int NP, bool PARTIAL>
__device__ __forceinline__ void path_finish(const Pack<NP> &Cp, const Pack<NP> &t, uint32_t ms, uint32_t P2s, bool active,
                                            Pack<NP> &Un, uint32_t &rmin)
{
    rmin = SGM_SENT;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        uint32_t v = pk_add(Cp.r[i], pk_min_s(pk_sub(t.r[i], ms), P2s));
        if (PARTIAL) v = active ? v : SGM_SENT;
        Un.r[i] = v;
        rmin = pk_min_s(rmin, v);
    }
}

// Ln - m on active lanes (ms = {m, m}, m the wave-uniform minimum); idle lanes keep the sentinel
template <int NP, bool PARTIAL>
__device__ __forceinline__ void path_normalise_splat(const Pack<NP> &Ln, uint32_t ms, bool active, Pack<NP> &out)
{
#pragma unroll
    for (int i = 0; i < NP; i++) {
        uint32_t v = pk_sub(Ln.r[i], ms);
        if (PARTIAL) v = active ? v : SGM_SENT;
        out.r[i] = v;
    }
}
template <int NP, bool PARTIAL>
__device__ __forceinline__ void path_normalise(const Pack<NP> &Ln, uint32_t m, bool active, Pack<NP> &out)
{
    path_normalise_splat<NP, PARTIAL>(Ln, splat16(m), active, out);
}

// Winner-take-all on the finished S of one pixel (A.6 steps 1-2; steps 3-4 run in k_select).
// Writes the record {reject ? ~0 : (minS << 16 | first best d), S[best-1] | S[best+1] << 16} to
// *rec: lane 0 stores the first word; the two halves of the second word are stored by whichever
// lanes hold disparities best-1 and best+1 (nobody, if they are outside [0, D) -- k_select reads
// them only for 0 < best < D-1).
//
// Uniqueness (upstream: reject if some d with |d - best| > 1 has S[d]*(100-uniq) < minS*100):
// for a positive weight that is "the smallest S outside {best-1, best, best+1}, times the
// weight, is below minS*100" -- one more packed wave reduction instead of per-element products
// and boolean mask arithmetic.  Non-positive weights (uniquenessRatio >= 100, POSW = false) take
// the literal per-element form.
// N pixels at a time (their reduction chains interleave; see wave_min_pk_n).  All stores come
// last so that both chains and the per-lane selects around them sit in one basic block.
// GW < 64: `lane` is the lane's index inside its group and every group has its own pixel (rec[n]
// differs between groups; the group's first lane stores the key word).
template <int NP, bool PARTIAL, bool POSW, int N, int GW = 64>
__device__ __forceinline__ void wta_pixels(const Pack<NP> (&Sn)[N], int lane, bool active, int D, int uniq,
                                           uint2 *const (&rec)[N])
{
    // packed constants in SGPRs: as inline operands they need op_sel_hi, and a VALU result
    // produced with op_sel costs its consumer a wait state on gfx950
    uint32_t c3, c15;
    asm("s_mov_b32 %0, 0x30003" : "=s"(c3));
    asm("s_mov_b32 %0, 0xf000f" : "=s"(c15));
    uint32_t key[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        uint32_t kmin = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t d0 = 2u * (NP * lane + i);
            const uint32_t klo = (Sn[n].r[i] << 16) | d0;
            const uint32_t khi = (Sn[n].r[i] & 0xffff0000u) | (d0 + 1u);
            kmin = min(kmin, min(klo, khi));
        }
        if (PARTIAL && !active) kmin = 0xffffffffu;
        key[n] = kmin;
    }
    group_min_u32_n<GW, N>(key);  // (minS << 16) | first best d
    const int wgt = 100 - uniq;
    uint32_t far[N], vm[N], vp[N];
    bool has_m[N], has_p[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        const int best = (int)(key[n] & 0xffffu);
        // e_i = d_lo(i) - (best - 1): the low half is best-1 / best / best+1 for e = 0 / 1 / 2,
        // the high half (d_lo + 1) for e = -1 / 0 / 1
        uint32_t f = SGM_SENT;
        vm[n] = vp[n] = 0;
        has_m[n] = has_p[n] = false;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int e = 2 * (NP * lane + i) - best + 1;
            const uint32_t v = Sn[n].r[i];
            if (e == 0 || e == -1) {
                vm[n] = e == 0 ? (v & 0xffffu) : (v >> 16);
                has_m[n] = true;
            }
            if (e == 2 || e == 1) {
                vp[n] = e == 2 ? (v & 0xffffu) : (v >> 16);
                has_p[n] = true;
            }
            if (POSW) {
                // per half: d - best + 1 in 0..2 <=> the disparity is one of best-1..best+1; those
                // halves are forced to 0x7fff (S <= 0x7fff, so OR does it) -- packed, no VCC
                const uint32_t u = pk_sub(splat16(2u * (NP * lane + i) + 1u) + 0x10000u, splat16(best));
                const uint32_t t = bits(as_s(pk_sub(pk_min_u(u, c3), c3)) >> as_s(c15));
                f = pk_min_s(f, (t & SGM_SENT) | v);
            }
        }
        if (PARTIAL) {
            has_m[n] = has_m[n] && active;
            has_p[n] = has_p[n] && active;
            if (!active) f = SGM_SENT;
        }
        far[n] = min(f & 0xffffu, f >> 16);  // one value per lane: the chain can use v_min_u32_dpp
    }
    if (POSW) group_min_u32_n<GW, N>(far);  // D >= 16: disparities outside best-1..best+1 always exist
#pragma unroll
    for (int n = 0; n < N; n++) {
        const int minS = (int)(key[n] >> 16), best = (int)(key[n] & 0xffffu);
        const int thr = minS * 100;
        bool reject;
        if (POSW) {  // wgt > 0 (the host picks the instantiation)
            reject = (int)far[n] * wgt < thr;
        } else {
            bool bad = false;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int d0 = 2 * (NP * lane + i);
                const int slo = (int)(Sn[n].r[i] & 0xffffu), shi = (int)(Sn[n].r[i] >> 16);
                bad |= (__mul24(slo, wgt) < thr) && (abs(best - d0) > 1);
                bad |= (__mul24(shi, wgt) < thr) && (abs(best - d0 - 1) > 1);
            }
            if (PARTIAL) bad = bad && active;
            if constexpr (GW == 64) {
                reject = __builtin_amdgcn_ballot_w64(bad) != 0ull;
            } else {  // "any lane of my group"
                uint32_t t[1] = {bad ? 0u : 1u};
                group_min_u32_n<GW, 1>(t);
                reject = t[0] == 0u;
            }
        }
        // all costs saturated: upstream keeps bestDisp = -1; the pixel ends invalid and never
        // wins a right-view slot (32767 > 32767 is false)
        reject = reject || (minS == SGM_MAX_COST);
        // The three values are materialised in VGPRs of their own under the full EXEC mask; only
        // the stores themselves run under a partial mask.  (gfx950, seen with NP = 4: when hipcc
        // sank these selects into the masked blocks it reused the registers of the S vector whose
        // buffer_store_dwordx4 had just been issued, and the stored S came out wrong in a few
        // lanes now and then -- tools/dbg_case.py, DESIGN.md 4.3 "masked writes after wide stores".)
        uint32_t kv = reject ? 0xffffffffu : key[n], m16 = vm[n], p16 = vp[n];
        asm volatile("" : "+v"(kv), "+v"(m16), "+v"(p16));
        uint16_t *nb = reinterpret_cast<uint16_t *>(&rec[n]->y);
        if (has_m[n]) nb[0] = (uint16_t)m16;
        if (has_p[n]) nb[1] = (uint16_t)p16;
        if (lane == 0) rec[n]->x = kv;
    }
}
template <int NP, bool PARTIAL, bool POSW>
__device__ __forceinline__ void wta_pixel(const Pack<NP> &Sn, int lane, bool active, int D, int uniq, uint2 *rec)
{
    const Pack<NP> S1[1] = {Sn};
    uint2 *const r1[1] = {rec};
    wta_pixels<NP, PARTIAL, POSW, 1>(S1, lane, active, D, uniq, r1);
}

#ifndef SGM_PREPASS_PB
#define SGM_PREPASS_PB 4
#endif

struct Cursor {
    int xi, y;
};

// Band-boundary state written by the PATH_BOUNDARY pre-pass and read by k_sweep:
//   bnd [band][x][3][D] int16 : normalised L of the three directions that come from the
//   previous row, slot 0: predecessor one step EARLIER in the sweep's x order, 1: same column,
//   2: one step LATER
struct Boundary {
    int16_t *L;
    int R;       // rows per band
    int slot;    // which of the three slots this launch writes
};

// one image row of a [H][W1][D] int16 volume as a buffer resource (no 4 GiB limit on the volume)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const int16_t *vol, int y, int W1, int D)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)(vol + (int64_t)y * W1 * D), 0, W1 * D * 2, 0x00020000);
}

// One direction, one wavefront per path line.  MODE: PATH_FIRST (S = L), PATH_ACCUM (S += L),
// PATH_LAST (S += L, winner-take-all, S stored only if keepS), PATH_BOUNDARY (no S; the state at
// band boundaries goes to bd.L; grid.y selects the role).  POSW: uniquenessRatio < 100 (the WTA
// variant without per-element products).
//
// Prefetch discipline as in k_sweep: buffer loads/stores (constant per-lane offset register +
// scalar offset inside the row), and iterations of two blocks in which the line neither ends
// nor wraps around the image border are straight-line code.
template <int NP, bool PARTIAL, int MODE, bool POSW>
__global__ __launch_bounds__(64) void k_path(Geom g, int rx, int ry, const int16_t *__restrict__ C,
                                             int16_t *__restrict__ S, int keepS,
                                             uint2 *__restrict__ wta, Boundary bd)
{
    constexpr int PB = 8;  // steps per prefetch block (two blocks in flight)
    const int lane = threadIdx.x;
    const int line = blockIdx.x;
    const int W1 = g.W1, D = g.D;
    if (MODE == PATH_BOUNDARY) {
        // all three roles in one launch (blockIdx.y): rx = +xdir, 0, -xdir where rx carries xdir
        bd.slot = blockIdx.y;
        rx = blockIdx.y == 0 ? rx : (blockIdx.y == 1 ? 0 : -rx);
    }
    const int nsteps = ry == 0 ? W1 : g.H;
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const int voff = lane_off * 2;
    const int pxb = D * 2;  // bytes per pixel
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const uint32_t init = active ? 0u : SGM_SENT;  // idle lanes act as the d = D sentinel
    constexpr bool READS_S = (MODE == PATH_ACCUM || MODE == PATH_LAST);

    Cursor ld, cp;  // load cursor runs ahead of the compute cursor
    if (ry == 0) {
        ld.y = line;
        ld.xi = rx > 0 ? 0 : W1 - 1;
    } else {
        ld.y = ry > 0 ? 0 : g.H - 1;
        ld.xi = line;
    }
    cp = ld;

    auto wrap = [&](Cursor &c) -> bool {  // true when the line left the image and re-enters
        if (c.xi >= W1) {
            c.xi = 0;
            return true;
        }
        if (c.xi < 0) {
            c.xi = W1 - 1;
            return true;
        }
        return false;
    };
    auto stays = [&](const Cursor &c, int n) { const int xe = c.xi + (n - 1) * rx; return xe >= 0 && xe < W1; };

    Pack<NP> L;  // normalised state L_r(q,.) - min (all-zero when q is outside the domain)
    L.fill(init);
    ShiftRegs sr;
    Pack<NP> cA[PB], cB[PB], sA[PB], sB[PB];
    uint32_t hm = 0;  // headroom record: largest min_d L_r(p, d) along this line
    int to_boundary = bd.R - 1, next_band = 1;  // PATH_BOUNDARY bookkeeping

    auto load_fast = [&](Pack<NP> *cb, Pack<NP> *sb) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            const int y = ld.y + k * ry, so = (ld.xi + k * rx) * pxb;
            buf_load<NP>(cb[k], row_rsrc(C, y, W1, D), voff, so);
            if (READS_S) buf_load<NP>(sb[k], row_rsrc(S, y, W1, D), voff, so);
        }
        ld.xi += PB * rx;
        ld.y += PB * ry;
    };
    auto load_slow = [&](Pack<NP> *cb, Pack<NP> *sb, int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            if (step0 + k < nsteps) {
                buf_load<NP>(cb[k], row_rsrc(C, ld.y, W1, D), voff, ld.xi * pxb);
                if (READS_S) buf_load<NP>(sb[k], row_rsrc(S, ld.y, W1, D), voff, ld.xi * pxb);
                ld.xi += rx;
                ld.y += ry;
                wrap(ld);
            }
        }
    };

    auto one_step = [&](const Pack<NP> &cv, const Pack<NP> &sv, bool more) {
        Pack<NP> Ln, Lnorm;
        uint32_t rmin;
        path_elem<NP, PARTIAL>(cv, L, P1s, P2s, active, Ln, rmin, sr);
        const uint32_t mLs = wave_min1_splat(rmin);
        hm = max(hm, mLs & 0xffffu);
        path_normalise_splat<NP, PARTIAL>(Ln, mLs, active, Lnorm);
        if (MODE == PATH_BOUNDARY) {
            // state of the last row of a band, consumed by the first row of the next band
            if (to_boundary == 0) {
                const int64_t px = (int64_t)next_band * W1 + cp.xi;
                if (active && more) Lnorm.store(bd.L + (px * 3 + bd.slot) * D + lane_off);
                to_boundary = bd.R;
                next_band++;
            }
            to_boundary--;
        } else {
            Pack<NP> Sn;
#pragma unroll
            for (int i = 0; i < NP; i++) Sn.r[i] = MODE == PATH_FIRST ? Ln.r[i] : pk_adds_s(sv.r[i], Ln.r[i]);
            if (MODE != PATH_LAST || keepS) {
                if (active) buf_store<NP>(Sn, row_rsrc(S, cp.y, W1, D), voff, cp.xi * pxb);
            }
            if (MODE == PATH_LAST) {
                wta_pixel<NP, PARTIAL, POSW>(Sn, lane, active, D, g.uniq, wta + ((int64_t)cp.y * g.W + g.minX1 + cp.xi));
            }
        }
        L = Lnorm;
    };
    auto compute_fast = [&](Pack<NP> *cb, Pack<NP> *sb, int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            one_step(cb[k], sb[k], step0 + k + 1 < nsteps);
            cp.xi += rx;
            cp.y += ry;
        }
    };
    auto compute_slow = [&](Pack<NP> *cb, Pack<NP> *sb, int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            if (step0 + k < nsteps) {
                one_step(cb[k], sb[k], step0 + k + 1 < nsteps);
                cp.xi += rx;
                cp.y += ry;
                if (wrap(cp)) L.fill(init);
            }
        }
    };

    if (PB <= nsteps && stays(ld, PB)) {
        load_fast(cA, sA);
        wrap(ld);
    } else {
        load_slow(cA, sA, 0);
    }
    for (int s0 = 0; s0 < nsteps; s0 += 2 * PB) {
        const bool fast = s0 + 3 * PB <= nsteps && stays(ld, 2 * PB) && stays(cp, 2 * PB);
        if (fast) {
            load_fast(cB, sB);
            compute_fast(cA, sA, s0);
            load_fast(cA, sA);
            compute_fast(cB, sB, s0 + PB);
            wrap(ld);  // a cursor may stand exactly one past the border now
            if (wrap(cp)) L.fill(init);
        } else {
            load_slow(cB, sB, s0 + PB);
            compute_slow(cA, sA, s0);
            load_slow(cA, sA, s0 + 2 * PB);
            compute_slow(cB, sB, s0 + PB);
        }
    }
    if (g.hr && lane == 0) headroom_raise(g.hr + 1, hm);
}

// ------------------------------------------------------------------------------------------
// Boundary pre-pass for k_sweep, all three roles in one wave: one line of each of the directions
// (rx = +xdir, 0, -xdir; ry = ydir) is advanced together, one image row per step.  Three
// independent dependency chains per wave (the scheduler interleaves them).  Only the normalised
// state at band boundaries is stored (Boundary layout above).
//
// C traffic.  A pixel of C is needed by three lines (one per role).  With a fixed line per wave the
// three readers of a pixel drift apart as the diagonals move (they sit on different XCDs, whose L2s
// share nothing): C was fetched three times per pass.  Now the image is walked in CHUNKS of rows
// (one launch per chunk, steps [s_begin, s_end)), and within a chunk a wave follows the three
// lines that cross ITS base column b in the middle of the chunk.  Workgroups are dealt to the
// XCDs round-robin (MI355X_MICROARCH.md: blocks b and b + 8 share an XCD -- observed, not a
// contract; it only decides speed), so base columns are laid out per XCD group: group i of
// blockIdx % 8 owns columns [i*cpx, (i+1)*cpx).  The three roles of a group then read the same
// C rows within +-chunk/2 columns of each other at about the same time: two of the three reads hit
// the group's L2.  Between chunks the state of every line goes through a small ping-pong buffer
// indexed by (role, column at the chunk border): state_in / state_out, [3][W1][D] int16 each.
//
// Prefetch discipline (see kernels_sweep.h): the cost volume is one buffer resource, every load
// is "constant per-lane offset register + scalar byte offset", and an iteration of two blocks
// in which no diagonal leaves the image is straight-line code, so hipcc keeps the next block's
// loads in flight (counted vmcnt).  The rare iterations with a wrap take the per-step path.
// Byte offsets are 32-bit (the host checks volume bytes < 2^31).
#ifndef SGM_PREPASS_WPB
#define SGM_PREPASS_WPB 4  // waves (adjacent base columns) per workgroup: a quarter of the workgroups to dispatch per chunk
#endif
template <int NP, bool PARTIAL, int PB = SGM_PREPASS_PB>
__global__ __launch_bounds__(64 * SGM_PREPASS_WPB) __attribute__((amdgpu_waves_per_eu(NP <= 2 ? (PB <= 2 ? 5 : 4) : 2)))
void k_prepass3(Geom g, int xdir, int ydir, const int16_t *__restrict__ C,
                                                 int16_t *__restrict__ bndL, int R, int s_begin, int s_end,
                                                 const int16_t *__restrict__ state_in, int16_t *__restrict__ state_out,
                                                 int cpx)
{
    // PB = rows per prefetch block: 2 blocks x 3 roles in registers.  PB = 4 keeps 4 waves per SIMD (a
    // pre-pass that has the GPU to itself); PB = 2 fits under 96 registers for the pass that shares the
    // SIMDs with the sweep's waves (its occupancy there is decided by what those leave over).
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W1 = g.W1, D = g.D, H = g.H;
    // base column of this wave: XCD-group layout (cpx > 0) or the plain one
    int base = blockIdx.x * SGM_PREPASS_WPB + wave;
    if (cpx > 0) {
        const int grp = blockIdx.x & 7, slot = (blockIdx.x >> 3) * SGM_PREPASS_WPB + wave;
        base = grp * cpx + slot;
        if (slot >= cpx) return;
    }
    if (base >= W1) return;
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    const uint32_t init = active ? 0u : SGM_SENT;
    const int rx[3] = {xdir, 0, -xdir};
    const uint32_t row_bytes = (uint32_t)W1 * D * 2;
    const __amdgpu_buffer_rsrc_t Cbuf =
        __builtin_amdgcn_make_buffer_rsrc((void *)C, 0, (int)(row_bytes * (uint32_t)H), 0x00020000);
    const int voff = lane_off * 2;
    uint32_t stride[3];  // byte stride of one step of each role (two's complement for negative steps)
#pragma unroll
    for (int d = 0; d < 3; d++) stride[d] = (uint32_t)(ydir * (int)row_bytes + rx[d] * D * 2);

    // per role: column of the next pixel of the load / compute cursors, byte offset of the load cursor.
    // The line of role d stands at column base in the middle row of the chunk.
    int xl[3], xc[3];
    uint32_t offl[3];
    const int y_begin = ydir > 0 ? s_begin : H - 1 - s_begin;
    const int back = (s_begin + s_end) / 2 - s_begin;  // steps from the chunk's first row to its middle
    Pack<NP> L[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int x = (base - rx[d] * back) % W1;  // |rx * back| may exceed W1 on narrow frames
        if (x < 0) x += W1;
        xl[d] = xc[d] = x;
        offl[d] = ((uint32_t)y_begin * W1 + x) * D * 2;
        L[d].fill(init);
        if (s_begin > 0 && active) L[d].load(state_in + ((int64_t)d * W1 + x) * D + lane_off);
    }
    ShiftRegs sr[3];
    Pack<NP> cA[PB][3], cB[PB][3];

    // does a diagonal that stands at column x stay inside [0, W1) for the next n steps ?
    auto stays = [&](int x, int d, int n) { const int xe = x + (n - 1) * rx[d]; return xe >= 0 && xe < W1; };
    auto wrap_load = [&]() {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (xl[d] >= W1) {
                xl[d] -= W1;
                offl[d] -= row_bytes;
            } else if (xl[d] < 0) {
                xl[d] += W1;
                offl[d] += row_bytes;
            }
        }
    };
    auto wrap_compute = [&]() {  // a line that left the image restarts from the zero state
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (xc[d] >= W1) {
                xc[d] = 0;
                L[d].fill(init);
            } else if (xc[d] < 0) {
                xc[d] = W1 - 1;
                L[d].fill(init);
            }
        }
    };

    auto load_fast = [&](Pack<NP>(*cb)[3]) {  // PB rows, no wrap inside
#pragma unroll
        for (int k = 0; k < PB; k++)
#pragma unroll
            for (int d = 0; d < 3; d++) buf_load<NP>(cb[k][d], Cbuf, voff, (int)(offl[d] + (uint32_t)k * stride[d]));
#pragma unroll
        for (int d = 0; d < 3; d++) {
            xl[d] += PB * rx[d];
            offl[d] += (uint32_t)PB * stride[d];
        }
    };
    auto load_slow = [&](Pack<NP>(*cb)[3], int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            if (step0 + k < s_end) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    buf_load<NP>(cb[k][d], Cbuf, voff, (int)offl[d]);
                    xl[d] += rx[d];
                    offl[d] += stride[d];
                }
                wrap_load();
            }
        }
    };

    // rows s with (s + 1) % R == 0 end a band: their state goes to the record of band (s + 1) / R
    int to_boundary = R - 1 - s_begin % R, next_band = s_begin / R + 1;
    const int bnd_row = W1 * 3 * D;  // int16 elements of one band's record
    uint32_t hm = 0;
    auto one_step = [&](Pack<NP> *c3, bool store_ok) {
        Pack<NP> N[3];
        uint32_t r[3];
#pragma unroll
        for (int d = 0; d < 3; d++) path_elem<NP, PARTIAL>(c3[d], L[d], P1s, P2s, active, N[d], r[d], sr[d]);
        uint32_t ms[3];
        wave_min3_splat(r[0], r[1], r[2], ms);
        hm = max(hm, max(ms[0], max(ms[1], ms[2])));  // headroom record (splats: see k_sweep)
#pragma unroll
        for (int d = 0; d < 3; d++) path_normalise_splat<NP, PARTIAL>(N[d], ms[d], active, L[d]);
        if (to_boundary == 0) {
            if (store_ok) {
                // the band's record [x][3][D] as a buffer resource: scalar address arithmetic only, no
                // 64-bit per-lane pointers (they cost the registers that decide 4 or 3 waves per SIMD)
                const __amdgpu_buffer_rsrc_t brec = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(bndL + (int64_t)next_band * bnd_row), 0, (int)((uint32_t)bnd_row * 2u), 0x00020000);
#pragma unroll
                for (int d = 0; d < 3; d++)
                    if (!PARTIAL || active) buf_store<NP>(L[d], brec, voff, (xc[d] * 3 + d) * D * 2);
            }
            to_boundary = R;
            next_band++;
        }
        to_boundary--;
    };
    auto compute_fast = [&](Pack<NP>(*cb)[3], int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            one_step(cb[k], step0 + k + 1 < H);
#pragma unroll
            for (int d = 0; d < 3; d++) xc[d] += rx[d];
        }
    };
    auto compute_slow = [&](Pack<NP>(*cb)[3], int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            if (step0 + k < s_end) {
                one_step(cb[k], step0 + k + 1 < H);
#pragma unroll
                for (int d = 0; d < 3; d++) xc[d] += rx[d];
                wrap_compute();
            }
        }
    };

    // block s0 is in cA (loaded here); each iteration loads s0+PB -> cB and s0+2PB -> cA
    if (s_begin + PB <= s_end && stays(xl[0], 0, PB) && stays(xl[2], 2, PB)) {
        load_fast(cA);
        wrap_load();
    } else {
        load_slow(cA, s_begin);
    }
    for (int s0 = s_begin; s0 < s_end; s0 += 2 * PB) {
        const bool fast = s0 + 3 * PB <= s_end && stays(xl[0], 0, 2 * PB) && stays(xl[2], 2, 2 * PB) &&
                          stays(xc[0], 0, 2 * PB) && stays(xc[2], 2, 2 * PB);
        // the chunk's last two blocks (chunk heights are multiples of 2 * PB wherever H allows): the same
        // straight-line code without the second load -- a chunk ends without a drained pipeline
        const bool last = s0 + 2 * PB == s_end && stays(xl[0], 0, PB) && stays(xl[2], 2, PB) &&
                          stays(xc[0], 0, 2 * PB) && stays(xc[2], 2, 2 * PB);
        if (fast) {
            load_fast(cB);
            compute_fast(cA, s0);
            load_fast(cA);
            compute_fast(cB, s0 + PB);
            wrap_load();      // a cursor may stand exactly one past the border now
            wrap_compute();
        } else if (last) {
            load_fast(cB);
            compute_fast(cA, s0);
            compute_fast(cB, s0 + PB);
            wrap_compute();
        } else {
            load_slow(cB, s0 + PB);
            compute_slow(cA, s0);
            load_slow(cA, s0 + 2 * PB);
            compute_slow(cB, s0 + PB);
        }
    }
    // hand the three lines to whichever waves pick them up in the next chunk
    if (s_end < H && active) {
#pragma unroll
        for (int d = 0; d < 3; d++) L[d].store(state_out + ((int64_t)d * W1 + xc[d]) * D + lane_off);
    }
    if (g.hr && lane == 0) headroom_raise(g.hr + 1, hm & 0xffffu);
}

}  // namespace sgm

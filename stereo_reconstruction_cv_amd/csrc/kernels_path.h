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

// One step of the recurrence for the D disparities a wavefront holds.  Lq / mq: predecessor
// state (idle lanes of a partial wave hold the MAX_COST sentinel); returns L(p, .) and its min.
template <int NP, bool PARTIAL>
__device__ __forceinline__ void path_recur(const Pack<NP> &Cp, const Pack<NP> &Lq, uint32_t mq, uint32_t P1s,
                                           uint32_t P2, bool active, Pack<NP> &Ln, uint32_t &mn)
{
    const uint32_t up = from_lower_lane(Lq.r[NP - 1], SGM_SENT);
    const uint32_t dn = from_upper_lane(Lq.r[0], SGM_SENT);
    const uint32_t mP2s = splat16(mq + P2), ms = splat16(mq);
    uint32_t rmin = SGM_SENT;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint32_t prevp = i == 0 ? up : Lq.r[i - 1];
        const uint32_t nextp = i == NP - 1 ? dn : Lq.r[i + 1];
        const uint32_t lm1 = __builtin_amdgcn_alignbit(Lq.r[i], prevp, 16);
        const uint32_t lp1 = __builtin_amdgcn_alignbit(nextp, Lq.r[i], 16);
        uint32_t t = pk_adds_s(pk_min_s(lm1, lp1), P1s);
        t = pk_min_s(pk_min_s(t, Lq.r[i]), mP2s);
        uint32_t v = pk_add(Cp.r[i], pk_sub(t, ms));
        if (PARTIAL) v = active ? v : SGM_SENT;
        Ln.r[i] = v;
        rmin = pk_min_s(rmin, v);
    }
    mn = wave_min_u32(min(rmin & 0xffffu, rmin >> 16));
}

// Winner-take-all on the finished S of one pixel (A.6 steps 1-2; steps 3-4 run in k_select):
// returns the record {reject ? ~0 : (minS << 16 | first best d), S[best-1] | S[best+1] << 16}.
template <int NP, bool PARTIAL>
__device__ __forceinline__ uint2 wta_pixel(const Pack<NP> &Sn, int lane, bool active, int D, int uniq)
{
    uint32_t kmin = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const uint32_t d0 = 2u * (NP * lane + i);
        const uint32_t klo = (Sn.r[i] << 16) | d0;
        const uint32_t khi = (Sn.r[i] & 0xffff0000u) | (d0 + 1u);
        kmin = min(kmin, min(klo, khi));
    }
    if (PARTIAL && !active) kmin = 0xffffffffu;
    const uint32_t key = wave_min_u32(kmin);  // (minS << 16) | first best d
    const int minS = (int)(key >> 16), best = (int)(key & 0xffffu);
    const int thr = minS * 100, wgt = 100 - uniq;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int d0 = 2 * (NP * lane + i);
        const int slo = (int)(Sn.r[i] & 0xffffu), shi = (int)(Sn.r[i] >> 16);
        bad |= (slo * wgt < thr) && (abs(best - d0) > 1);
        bad |= (shi * wgt < thr) && (abs(best - d0 - 1) > 1);
    }
    if (PARTIAL) bad = bad && active;
    bool reject = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    // all costs saturated: upstream keeps bestDisp = -1; the pixel ends invalid and never wins
    // a right-view slot (32767 > 32767 is false)
    reject = reject || (minS == SGM_MAX_COST);
    auto fetch = [&](int d) -> uint32_t {
        const int p = d >> 1, ln = p / NP, i = p - ln * NP;
        uint32_t v = Sn.r[0];
#pragma unroll
        for (int q = 1; q < NP; q++) v = (i == q) ? Sn.r[q] : v;
        v = __builtin_amdgcn_readlane(v, ln);
        return (d & 1) ? (v >> 16) : (v & 0xffffu);
    };
    uint32_t nb = 0;
    if (best > 0 && best < D - 1) nb = fetch(best - 1) | (fetch(best + 1) << 16);
    return make_uint2(reject ? 0xffffffffu : key, nb);
}

struct Cursor {
    int xi, y;
};

// Band-boundary state written by the PATH_BOUNDARY pre-pass and read by k_sweep:
//   bnd [band][x][3][D] int16 : L of directions (x-1), (x), (x+1) of the previous row
//   bmin[band][x][4]    int32 : their minima
struct Boundary {
    int16_t *L;
    int32_t *M;
    int R;       // rows per band
    int slot;    // which of the three directions this launch writes (0: rx=+1, 1: rx=0, 2: rx=-1)
};

template <int NP, bool PARTIAL, int MODE>
__global__ __launch_bounds__(64) void k_path(Geom g, int rx, int ry, const int16_t *__restrict__ C,
                                             int16_t *__restrict__ S, int keepS,
                                             uint2 *__restrict__ wta, Boundary bd)
{
    constexpr int PB = 8;  // steps per prefetch block
    const int lane = threadIdx.x;
    const int line = blockIdx.x;
    const int W1 = g.W1, D = g.D;
    const int nsteps = ry == 0 ? W1 : g.H;
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;  // idle lanes load lane 0's data (ignored)
    const uint32_t P1s = splat16((uint32_t)g.P1);
    const uint32_t init = active ? 0u : SGM_SENT;  // idle lanes act as the d = D sentinel

    Cursor ld, cp;  // load cursor runs ahead of the compute cursor
    if (ry == 0) {
        ld.y = line;
        ld.xi = rx > 0 ? 0 : W1 - 1;
    } else {
        ld.y = ry > 0 ? 0 : g.H - 1;
        ld.xi = line;
    }
    cp = ld;

    auto advance = [&](Cursor &c) -> bool {  // returns true when the predecessor left the domain
        c.xi += rx;
        c.y += ry;
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

    Pack<NP> L;
    L.fill(init);
    uint32_t m = 0;

    Pack<NP> cA[PB], cB[PB], sA[PB], sB[PB];
    constexpr bool READS_S = (MODE == PATH_ACCUM || MODE == PATH_LAST);

    auto load_block = [&](Pack<NP> *cb, Pack<NP> *sb, int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            if (step0 + k < nsteps) {
                const int64_t off = ((int64_t)ld.y * W1 + ld.xi) * D + lane_off;
                cb[k].load(C + off);
                if (READS_S) sb[k].load(S + off);
                advance(ld);
            }
        }
    };

    auto compute_block = [&](Pack<NP> *cb, Pack<NP> *sb, int step0) {
#pragma unroll
        for (int k = 0; k < PB; k++) {
            if (step0 + k < nsteps) {
                Pack<NP> Ln;
                uint32_t m_new;
                path_recur<NP, PARTIAL>(cb[k], L, m, P1s, (uint32_t)g.P2, active, Ln, m_new);

                if (MODE == PATH_BOUNDARY) {
                    // state of the last row of a band, consumed by the first row of the next band
                    const int j = ry > 0 ? cp.y : g.H - 1 - cp.y;  // row index in sweep order
                    if ((j + 1) % bd.R == 0 && j + 1 < g.H) {
                        const int64_t px = (int64_t)((j + 1) / bd.R) * W1 + cp.xi;
                        if (active) Ln.store(bd.L + (px * 3 + bd.slot) * D + lane_off);
                        if (lane == 0) bd.M[px * 4 + bd.slot] = (int32_t)m_new;
                    }
                } else {
                    const int64_t off = ((int64_t)cp.y * W1 + cp.xi) * D + lane_off;
                    Pack<NP> Sn;
#pragma unroll
                    for (int i = 0; i < NP; i++)
                        Sn.r[i] = MODE == PATH_FIRST ? Ln.r[i] : pk_adds_s(sb[k].r[i], Ln.r[i]);
                    if (MODE != PATH_LAST || keepS) {
                        if (active) Sn.store(S + off);
                    }
                    if (MODE == PATH_LAST) {
                        const uint2 rec = wta_pixel<NP, PARTIAL>(Sn, lane, active, D, g.uniq);
                        if (lane == 0) wta[(int64_t)cp.y * g.W + cp.xi + g.minX1] = rec;
                    }
                }

                L = Ln;
                m = m_new;
                if (advance(cp)) {
                    L.fill(init);
                    m = 0;
                }
            }
        }
    };

    load_block(cA, sA, 0);
    for (int s0 = 0; s0 < nsteps; s0 += 2 * PB) {
        load_block(cB, sB, s0 + PB);
        compute_block(cA, sA, s0);
        load_block(cA, sA, s0 + 2 * PB);
        compute_block(cB, sB, s0 + PB);
    }
}

}  // namespace sgm

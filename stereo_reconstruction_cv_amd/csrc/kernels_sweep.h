// kernels_sweep.h -- four SGM directions fused in one sweep over the image.
//
// Replaces one whole pass of upstream computeDisparitySGBM (/root/reference/main.ipynb:668;
// SURVEY.md A.5): for every pixel the path from the previous pixel of the same row and the
// three paths from the previously processed row, S = sat(sum), and -- SWEEP_LAST only; the
// engine's default runs the winner-take-all as its own pass over S (k_wta_t) after a SWEEP_ACCUM --
// the winner-take-all scan.  Same arithmetic as k_path (path_elem / wta_pixels), different
// schedule: traffic per pass is "read C once, write S once" instead of 2-3 volumes per direction.
//
// Schedule.  The image is cut into bands of R rows (in sweep order).  One workgroup owns one
// band: R compute waves (one image row each) + 1 loader wave.  The compute waves run the row
// recurrence in lockstep, wave r two steps behind wave r-1, one workgroup barrier per step (a
// step is two pixels for D <= 256): the (normalised) state a row needs from the row above -- three L vectors per pixel --
// travels through a small LDS ring per wave and never touches HBM.  Bands do not wait for each
// other: the state of the row above a band comes from the boundary pre-pass (k_prepass3: three
// read-only line scans that store L only at band boundaries); the loader wave prefetches
// it from HBM into registers and feeds an LDS ring two steps ahead of wave 0.
//
// Everything is expressed in the sweep's own pixel order k (x = k or W1-1-k): role A = path
// whose predecessor is pixel k-1 of the row above, B = pixel k, C = pixel k+1; which image
// directions these are depends on the sweep direction and is decided on the host.
#pragma once
#include "kernels_path.h"
#include <type_traits>

namespace sgm {

enum { SWEEP_FIRST = 0, SWEEP_ACCUM = 1, SWEEP_LAST = 2 };

struct SweepArgs {
    int ydir, xdir;  // (+1,+1): rows top->bottom, x ascending; (-1,-1): second pass of MODE_HH
    int R;           // rows per band = compute waves per workgroup
    const int16_t *C;
    int16_t *S;
    const int16_t *bndL;  // [band][x][3 roles][D], normalised
    uint2 *wta;
    int keepS;
    int dbg;  // timing experiments only (results become wrong): 64 = loader wave skips its HBM loads
    int band0;  // the launch covers bands band0 .. band0 + gridDim.x - 1 (row-chunk pipelining of the first pass)
};

constexpr int SWEEP_MAX_ROWS = 11;  // compute waves per workgroup (768 threads = 3 waves per SIMD -> 168 VGPRs per lane)
// pixels a wave advances per lockstep step (one barrier per step): two give the scheduler two
// independent dependency chains per wave and halve the barrier / LDS round trips per pixel
__host__ __device__ constexpr int sweep_pps(int NP) { return NP == 4 ? 1 : (NP == 1 ? 4 : 2); }
// pixels of hand-off state kept per producer: consumers run 2 steps behind their producer
__host__ __device__ constexpr int sweep_ring(int NP) { return 4 * sweep_pps(NP); }

__host__ __device__ constexpr int sweep_slot_dwords(int NP) { return 192 * NP; }
static inline size_t sweep_lds_bytes(int NP, int R)
{
    return (size_t)(R + 1) * sweep_ring(NP) * sweep_slot_dwords(NP) * 4;
}

__device__ __forceinline__ void wg_barrier()
{
    // LDS traffic of this step must have landed before the other waves pass the barrier;
    // global loads/stores in flight (prefetch, S stores) are deliberately NOT drained
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NP>
__device__ __forceinline__ void lds_load(Pack<NP> &p, const uint32_t *src)
{
    typename PackVec<NP>::type v = *reinterpret_cast<const typename PackVec<NP>::type *>(src);
    __builtin_memcpy(p.r, &v, sizeof(v));
}
template <int NP>
__device__ __forceinline__ void lds_store(const Pack<NP> &p, uint32_t *dst)
{
    typename PackVec<NP>::type v;
    __builtin_memcpy(&v, p.r, sizeof(v));
    *reinterpret_cast<typename PackVec<NP>::type *>(dst) = v;
}

// Every byte the sweep loads (C, S, boundary state) is used once: loaded "nt", so that these lines leave
// the L2s first -- the upward pre-pass that runs beside the downward sweep keeps its C lines for its second
// and third reader (4K MODE_HH: frame 10.62 -> 10.48 ms on one box, A/B through SGM_HIP_LIB).
#ifndef SGM_NT_SWEEP_LOADS
#define SGM_NT_SWEEP_LOADS 1
#endif
template <int NP, bool PARTIAL, int MODE, bool POSW>
__global__ __launch_bounds__(SWEEP_MAX_ROWS * 64 + 64) void k_sweep(Geom g, SweepArgs a)
{
    constexpr int LDAUX = SGM_NT_SWEEP_LOADS ? 2 : 0;  // "nt": every byte the sweep loads is used once
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int PPS = sweep_pps(NP);
    constexpr int RING = sweep_ring(NP);
    constexpr int PB = RING;  // prefetch block (pixels) = ring depth, so slot(k) = k % PB is static
    constexpr int SLOT = sweep_slot_dwords(NP);
    constexpr int ROLE = 64 * NP;  // dwords per role inside a slot
    const int R = a.R;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int band = a.band0 + blockIdx.x;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int T = (W1 + PPS - 1) / PPS + 2 * (R - 1);  // lockstep steps
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const uint32_t init = active ? 0u : SGM_SENT;
    // ring 0 is fed by the loader (row above the band), ring r+1 by compute wave r
    uint32_t *const ring0 = lds + lane * NP;

    // Out-of-image neighbours are ordinary ring slots holding the start state: "pixel -1" lives in
    // slot RING-1 (written before the prologue barrier, not reused until pixel RING-1 exists) and
    // "pixel W1" is written by each producer one step after its last real pixel.  The per-pixel
    // code therefore needs no border branches.
    auto write_start_state = [&](uint32_t *ring, int slot) {
        Pack<NP> v;
        v.fill(init);
#pragma unroll
        for (int d = 0; d < 3; d++) lds_store<NP>(v, ring + slot * SLOT + d * ROLE);
    };

    if (wave == R) {
        // ==== loader wave: boundary state HBM -> registers -> LDS, 2 steps ahead of wave 0 ====
        const bool has_prev = band > 0 && !(a.dbg & 64);
        Pack<NP> bA[PB][3], bB[PB][3];
        // this band's boundary row as a buffer resource: [x][3 roles][D] int16
        const int row_bytes = W1 * 3 * D * 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.bndL + (int64_t)band * W1 * 3 * D), 0, row_bytes, 0x00020000);
        const int voff = lane_off * 2;
        const int px_bytes = 3 * D * 2;
        const int pk = a.xdir > 0 ? px_bytes : -px_bytes;           // byte step per pixel of the sweep order
        const int p0 = a.xdir > 0 ? 0 : (W1 - 1) * px_bytes;
        // FULL = every pixel of the block exists: no guards, so hipcc can count the loads in flight
        // (with a branch between issue and use it falls back to vmcnt(0) and the whole lockstep
        // workgroup waits for HBM latency every block)
        auto lb_t = [&](auto full_c, Pack<NP>(*b)[3], int k0) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int u = 0; u < PB; u++) {
                if (FULL || k0 + u < W1) {
                    const int so = p0 + (k0 + u) * pk;
#pragma unroll
                    for (int d = 0; d < 3; d++) buf_load<NP, LDAUX>(b[u][d], rsrc, voff, so + d * D * 2);
                }
            }
        };
        auto wb_t = [&](auto full_c, Pack<NP>(*b)[3], int u, int k) {  // pixel k, k % PB == u
            constexpr bool FULL = decltype(full_c)::value;
            if (!FULL && k > W1) return;
            uint32_t *slot = ring0 + u * SLOT;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                Pack<NP> v;
                if (FULL || k < W1) {
                    v = b[u][d];
                    if (PARTIAL && !active) v.fill(SGM_SENT);
                } else {
                    v.fill(init);  // the virtual pixel W1
                }
                lds_store<NP>(v, slot + d * ROLE);
            }
        };
        int t = 0;  // lockstep steps done; step t writes pixels PPS*(t+2) .. PPS*(t+2)+PPS-1
        if (!has_prev) {
            // first band of the sweep: the row above is the all-zero start state everywhere
#pragma unroll
            for (int u = 0; u < RING; u++) write_start_state(ring0, u);
            wg_barrier();
            for (; t < T; t++) wg_barrier();
            return;
        }
        write_start_state(ring0, RING - 1);
        const std::true_type full{};
        const std::false_type part{};
        // prologue: pixels 0 .. 2*PPS-1, then the rest of block 0
        if (2 * PB <= W1) {
            lb_t(full, bA, 0);
            lb_t(full, bB, PB);
#pragma unroll
            for (int p = 0; p < 2 * PPS; p++) wb_t(full, bA, p, p);
            wg_barrier();
#pragma unroll
            for (int u0 = 2 * PPS; u0 < PB; u0 += PPS) {
#pragma unroll
                for (int p = 0; p < PPS; p++) wb_t(full, bA, u0 + p, u0 + p);
                wg_barrier();
                t++;
            }
        } else {
            lb_t(part, bA, 0);
            lb_t(part, bB, PB);
#pragma unroll
            for (int p = 0; p < 2 * PPS; p++) wb_t(part, bA, p, p);
            wg_barrier();
#pragma unroll
            for (int u0 = 2 * PPS; u0 < PB; u0 += PPS) {
                if (t < T) {
#pragma unroll
                    for (int p = 0; p < PPS; p++) wb_t(part, bA, u0 + p, u0 + p);
                    wg_barrier();
                    t++;
                }
            }
        }
        int k0 = PB;  // bB holds block [k0, k0+PB), bA is free
        // steady state: blocks k0 (in bB), k0+PB (to bA), k0+2PB (to bB) all full -> straight-line
        for (; k0 + 3 * PB <= W1; k0 += 2 * PB) {
            lb_t(full, bA, k0 + PB);
#pragma unroll
            for (int u0 = 0; u0 < PB; u0 += PPS) {
#pragma unroll
                for (int p = 0; p < PPS; p++) wb_t(full, bB, u0 + p, k0 + u0 + p);
                wg_barrier();
            }
            lb_t(full, bB, k0 + 2 * PB);
#pragma unroll
            for (int u0 = 0; u0 < PB; u0 += PPS) {
#pragma unroll
                for (int p = 0; p < PPS; p++) wb_t(full, bA, u0 + p, k0 + PB + u0 + p);
                wg_barrier();
            }
            t += 2 * (PB / PPS);
        }
        // tail: guarded blocks, then idle steps until every row has finished
        for (; t < T; k0 += 2 * PB) {
            lb_t(part, bA, k0 + PB);
#pragma unroll
            for (int u0 = 0; u0 < PB; u0 += PPS) {
                if (t < T) {
#pragma unroll
                    for (int p = 0; p < PPS; p++) wb_t(part, bB, u0 + p, k0 + u0 + p);
                    wg_barrier();
                    t++;
                }
            }
            lb_t(part, bB, k0 + 2 * PB);
#pragma unroll
            for (int u0 = 0; u0 < PB; u0 += PPS) {
                if (t < T) {
#pragma unroll
                    for (int p = 0; p < PPS; p++) wb_t(part, bA, u0 + p, k0 + PB + u0 + p);
                    wg_barrier();
                    t++;
                }
            }
        }
        return;
    }

    // ==== compute wave: one image row ==============================================================
    const int j = band * R + wave;  // row index in sweep order
    const int y = a.ydir > 0 ? j : H - 1 - j;
    const uint32_t *const prev = ring0 + wave * RING * SLOT;
    uint32_t *const mine = ring0 + (wave + 1) * RING * SLOT;
    write_start_state(mine, RING - 1);
    wg_barrier();  // prologue barrier
    if (j >= H) {  // row past the image (last band): keep the barrier count, do nothing
        for (int t = 0; t < T; t++) wg_barrier();
        return;
    }
    const uint32_t P1s = splat16((uint32_t)g.P1), P2s = splat16((uint32_t)g.P2);
    constexpr bool READS_S = MODE != SWEEP_FIRST;

    for (int i = 0; i < 2 * wave; i++) wg_barrier();  // start two steps behind the row above

    Pack<NP> L0;  // normalised state of the in-row path
    L0.fill(init);
    ShiftRegs sr0, srA, srB, srC;
    // headroom record (sgm_get_headroom): largest min_d L_r(p, d) of the row, all four directions.
    // The reductions return splats {m, m}, whose unsigned 32-bit maximum is the splat of the
    // maximum -- four scalar instructions per pixel, nothing added to the vector stream.
    uint32_t hm = 0;
    Pack<NP> cA[PB], cB[PB], sA[PB], sB[PB];
    // this row of C and S as buffer resources: one constant per-lane byte offset register plus a
    // scalar byte offset per pixel (b0 + k * bk), so no address VGPRs alias the load destinations
    const int row_bytes = W1 * D * 2;
    const __amdgpu_buffer_rsrc_t Crow = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.C + (int64_t)y * W1 * D), 0, row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t Srow = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.S + (int64_t)y * W1 * D), 0, row_bytes, 0x00020000);
    // stores of S go through a descriptor of their own: zero records (every store dropped by the
    // bounds check) when the last sweep need not keep S -- no branch inside the step
    const __amdgpu_buffer_rsrc_t Sst = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.S + (int64_t)y * W1 * D), 0, (MODE != SWEEP_LAST || a.keepS) ? row_bytes : 0, 0x00020000);
    const int voff = lane_off * 2;
    const int bk = a.xdir > 0 ? D * 2 : -D * 2;
    const int b0 = a.xdir > 0 ? 0 : (W1 - 1) * D * 2;
    const int wk = a.xdir > 0 ? 1 : -1;
    uint2 *const wrow = a.wta + (int64_t)y * g.W + g.minX1 + (a.xdir > 0 ? 0 : W1 - 1);

    // FULL blocks (all but the last of a row) are straight-line code without guards, so that the
    // scheduler can interleave the independent chains of the PPS pixels of a step
    auto load_block_t = [&](auto full_c, Pack<NP> *cb, Pack<NP> *sb, int k0) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int u = 0; u < PB; u++) {
            if (FULL || k0 + u < W1) {
                const int so = b0 + (k0 + u) * bk;
                buf_load<NP, LDAUX>(cb[u], Crow, voff, so);
                if (READS_S) buf_load<NP, LDAUX>(sb[u], Srow, voff, so);
            }
        }
    };
    // one pixel: four recurrences (minima reduced two directions at a time), hand-off, S, WTA
    auto pixel = [&](const Pack<NP> &Cp, const Pack<NP> &Sp, const Pack<NP> &QA, const Pack<NP> &QB,
                     const Pack<NP> &QC, int u, int k) {
        Pack<NP> N0, NA, NB, NC;
        uint32_t r0, rA, rB, rC;
        path_elem<NP, PARTIAL>(Cp, L0, P1s, P2s, active, N0, r0, sr0);
        path_elem<NP, PARTIAL>(Cp, QA, P1s, P2s, active, NA, rA, srA);
        path_elem<NP, PARTIAL>(Cp, QB, P1s, P2s, active, NB, rB, srB);
        path_elem<NP, PARTIAL>(Cp, QC, P1s, P2s, active, NC, rC, srC);
        uint32_t ms[4];  // {m, m} of the directions 0, A, B, C
        wave_min4_splat(r0, rA, rB, rC, ms);
        smax_u32(hm, ms[0]);
        smax_u32(hm, ms[1]);
        smax_u32(hm, ms[2]);
        smax_u32(hm, ms[3]);
        Pack<NP> LA, LB, LC;
        path_normalise_splat<NP, PARTIAL>(N0, ms[0], active, L0);
        path_normalise_splat<NP, PARTIAL>(NA, ms[1], active, LA);
        path_normalise_splat<NP, PARTIAL>(NB, ms[2], active, LB);
        path_normalise_splat<NP, PARTIAL>(NC, ms[3], active, LC);
        uint32_t *s = mine + u * SLOT;
        lds_store<NP>(LA, s + 0 * ROLE);
        lds_store<NP>(LB, s + 1 * ROLE);
        lds_store<NP>(LC, s + 2 * ROLE);
        Pack<NP> Sn;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            uint32_t v = pk_adds_s(pk_adds_s(N0.r[i], NA.r[i]), pk_adds_s(NB.r[i], NC.r[i]));
            if (READS_S) v = pk_adds_s(v, Sp.r[i]);
            Sn.r[i] = v;
        }
        if (!PARTIAL || active) buf_store<NP>(Sn, Sst, voff, b0 + k * bk);
        return Sn;
    };

    auto compute_block_t = [&](auto full_c, Pack<NP> *cb, Pack<NP> *sb, int k0) {  // k0 % PB == 0
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int u0 = 0; u0 < PB; u0 += PPS) {
            if (FULL || k0 + u0 < W1) {
                // normalised state of the row above for the PPS pixels of this step (LDS)
                Pack<NP> QA[PPS], QB[PPS], QC[PPS], Sn[PPS];
#pragma unroll
                for (int p = 0; p < PPS; p++) {
                    const int u = u0 + p;
                    lds_load<NP>(QA[p], prev + ((u + RING - 1) % RING) * SLOT + 0 * ROLE);
                    lds_load<NP>(QB[p], prev + u * SLOT + 1 * ROLE);
                    lds_load<NP>(QC[p], prev + ((u + 1) % RING) * SLOT + 2 * ROLE);
                }
#pragma unroll
                for (int p = 0; p < PPS; p++) {
                    const int u = u0 + p;
                    if (FULL || k0 + u < W1) Sn[p] = pixel(cb[u], sb[u], QA[p], QB[p], QC[p], u, k0 + u);
                }
                if (MODE == SWEEP_LAST) {  // winner-take-all of the step's pixels, chains interleaved
                    if (FULL || k0 + u0 + PPS <= W1) {
                        uint2 *recs[PPS];
#pragma unroll
                        for (int p = 0; p < PPS; p++) recs[p] = wrow + (k0 + u0 + p) * wk;
                        wta_pixels<NP, PARTIAL, POSW, PPS>(Sn, lane, active, D, g.uniq, recs);
                    } else {
#pragma unroll
                        for (int p = 0; p < PPS; p++)
                            if (k0 + u0 + p < W1)
                                wta_pixel<NP, PARTIAL, POSW>(Sn[p], lane, active, D, g.uniq, wrow + (k0 + u0 + p) * wk);
                    }
                }
                wg_barrier();
            }
        }
    };
    {
        const std::true_type full{};
        const std::false_type part{};
        int k0 = 0;
        if (PB <= W1) load_block_t(full, cA, sA, 0);
        else load_block_t(part, cA, sA, 0);
        // steady state: blocks k0 (in cA), k0+PB (to cB), k0+2PB (to cA) all full -> one straight-line
        // iteration, so the loads of the next block stay in flight across the current block
        for (; k0 + 3 * PB <= W1; k0 += 2 * PB) {
            load_block_t(full, cB, sB, k0 + PB);
            compute_block_t(full, cA, sA, k0);
            load_block_t(full, cA, sA, k0 + 2 * PB);
            compute_block_t(full, cB, sB, k0 + PB);
        }
        for (; k0 < W1; k0 += 2 * PB) {  // tail: guarded
            load_block_t(part, cB, sB, k0 + PB);
            compute_block_t(part, cA, sA, k0);
            load_block_t(part, cA, sA, k0 + 2 * PB);
            compute_block_t(part, cB, sB, k0 + PB);
        }
    }
    if (g.hr && lane == 0) headroom_raise(g.hr + 1, hm & 0xffffu);
    // one step after the last real pixel: the virtual pixel W1 (start state) for the row below
    if (wave < R - 1) {
        write_start_state(mine, W1 % RING);
        for (int i = 0; i < 2 * (R - 1 - wave); i++) wg_barrier();
    }
}

}  // namespace sgm

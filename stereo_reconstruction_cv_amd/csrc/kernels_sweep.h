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
    // chained schedule only (k_sweep_chain): bndL is then the record the bands hand to each other
    uint32_t *ctl;   // [0] ticket, [1 + f * nbands + b] = pixels of the last row of band b of frame f that have reached HBM;
                     // zeroed before every launch
    uint32_t *prog;  // (set by the kernel) progress words of the frame a workgroup is working on = ctl + 1 + f * nbands
    uint32_t *err;   // set when a bounded wait gave up; stays set until the host has reported it (sgm_engine.hip: check_chain)
    int nbands;
};

// Frames of one chained launch (sgm_pipeline_batch_device: several pairs per launch, so that the GPU is full although
// one frame's chain keeps only T / LAG workgroups busy).  Tickets go round the frames: ticket t = band t / nf of frame t % nf.
constexpr int CHAIN_MAX_FRAMES = 64;   // (the four pointer arrays of ChainFrames: 2 KB of the 4 KB a kernel's arguments may take)
struct ChainFrames {
    int nf;
    const int16_t *C[CHAIN_MAX_FRAMES];
    int16_t *S[CHAIN_MAX_FRAMES];
    int16_t *bnd[CHAIN_MAX_FRAMES];   // hand-off record [band][x][3][D] of each frame
    uint32_t *hr[CHAIN_MAX_FRAMES];   // headroom record of each frame
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
// ---- the waves of a band ---------------------------------------------------------------------------
// The body of a band is the same in the plain schedule (k_sweep: one workgroup per band, the state of
// the row above the band comes from the pre-pass) and in the chained one (k_sweep_chain: the state comes
// from the band above, through HBM, as that band produces it).  Every wave of a workgroup passes
// exactly 1 + T workgroup barriers per band, T = ceil(W1 / PPS) + 2 (R - 1): barrier 0 closes the
// prologue, barrier 1 + t closes lockstep step t.

template <int NP>
__device__ __forceinline__ void sweep_write_start_state(uint32_t *ring, int slot, uint32_t init)
{
    constexpr int SLOT = sweep_slot_dwords(NP), ROLE = 64 * NP;
    Pack<NP> v;
    v.fill(init);
#pragma unroll
    for (int d = 0; d < 3; d++) lds_store<NP>(v, ring + slot * SLOT + d * ROLE);
}

// Chained schedule: progress word of the band above (number of pixels of its last row, in sweep order,
// whose three path states are in HBM).  Polled by the loader wave only; relaxed agent-scope loads
// ("sc1": served behind this CU's L1), the payload is then read with sc1 loads as well
// (cdna_hip_programming.md Guideline 16, R1 with the measured sc1-load form: one lane of the producer
// stores the flag after its wave's payload stores are complete, the polling wave loads after its poll
// has matched).  Every wait is bounded: after CHAIN_GIVE_UP polls in a row that saw the word stand still it
// gives up, raises *err and lets the band run on (its results are then wrong and the host reports SGM_ERR_HIP
// at its next check: sgm_synchronize / sgm_check) -- the grid always drains.  The bound counts this wave's OWN
// polls (each a load that goes to L2 or HBM plus s_sleep: about a microsecond), not wall time: a queue that the
// hardware scheduler parks for a while (several processes time-slicing one GPU, a debugger) parks producer and
// consumer together and must not run the clock down, and a producer that still advances -- however slowly --
// restarts the count.
constexpr uint32_t CHAIN_GIVE_UP = 1u << 20;
struct ChainWait {
    const uint32_t *word = nullptr;  // progress word of the band above (nullptr: nothing to wait for)
    uint32_t *err = nullptr;
    uint32_t seen = 0;
    uint32_t pending = 0;  // a poll issued one prefetch block ago: by the time it is looked at, it has long returned
    __device__ __forceinline__ uint32_t peek() const
    {
        return __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    // Block boundary of the loader: take the poll issued a block ago, wait (rarely) for `need` pixels, issue the
    // next poll -- in front of the block's loads, so that it never returns later than data the wave waits for anyway.
    __device__ __forceinline__ void step(uint32_t need)
    {
        if (!word) return;
        seen = max(seen, (uint32_t)__builtin_amdgcn_readfirstlane(pending));
        until(need);
        pending = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void until(uint32_t need)
    {
        if (seen >= need) return;
        uint32_t idle = 0, last = seen;
        for (;;) {
            seen = peek();
            if (seen >= need) break;
            __builtin_amdgcn_s_sleep(8);
            if (seen != last) {  // the band above is alive
                last = seen;
                idle = 0;
            } else if (++idle > CHAIN_GIVE_UP) {
                if (lane_id() == 0) atomicOr(err, 1u);
                seen = 0x7fffffffu;  // this workgroup stops waiting for good
                break;
            }
        }
    }
};

// ==== loader wave: state of the row above the band, HBM -> registers -> LDS ring 0, ahead of wave 0 ====
// Blocks of LB pixels: the loads of block b + 1 are issued when the first pixel of block b goes to the ring, LB / PPS
// lockstep steps before their first use.  LB = ring depth (4 steps of prefetch distance) in both schedules.  For the
// chained schedule half-ring blocks (2 steps) were measured in round 3 -- every step a band runs less far ahead of what
// it needs is a step less of chain latency per band: a launch that is bound by that latency gained (16 pairs 1080p
// D=128: 6.65 -> 5.99 ms), the default batch lost (12 pairs 4K D=256: 23.6 -> 25.5 and 29.7 -> 30.4 ms per pass): with
// the GPU saturated the record's loads take longer than two steps to come back and the whole lockstep workgroup waits
// for them.  SGM_CHAIN_HALF_BLOCKS=1 builds that variant.
#ifndef SGM_CHAIN_HALF_BLOCKS
#define SGM_CHAIN_HALF_BLOCKS 0
#endif
template <int NP, bool PARTIAL, bool CHAIN>
__device__ __forceinline__ void sweep_loader_wave(const Geom &g, const SweepArgs &a, int band, int lane, uint32_t *lds)
{
    // plain schedule: every byte is used once -> "nt"; chained: the record was written by another CU a
    // moment ago -> "sc1" (agent scope: not from this CU's L1)
    constexpr int LDAUX = CHAIN ? 16 : (SGM_NT_SWEEP_LOADS ? 2 : 0);
    constexpr int PPS = sweep_pps(NP);
    constexpr int RING = sweep_ring(NP);
    constexpr int LB = (CHAIN && SGM_CHAIN_HALF_BLOCKS) ? RING / 2 : RING;  // pixels per prefetch block
    // first ring slot of a block: slot(k) = k % RING is static because blocks alternate between the two register
    // buffers: bA holds the blocks that start at multiples of 2 LB, bB the others
    constexpr int OFF_A = 0, OFF_B = LB % RING;
    static_assert(2 * PPS <= LB && LB % PPS == 0 && RING % LB == 0, "prologue writes 2 * PPS pixels of block 0");
    constexpr int SLOT = sweep_slot_dwords(NP);
    constexpr int ROLE = 64 * NP;  // dwords per role inside a slot
    const int R = a.R;
    const int W1 = g.W1, D = g.D;
    const int T = (W1 + PPS - 1) / PPS + 2 * (R - 1);  // lockstep steps
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const uint32_t init = active ? 0u : SGM_SENT;
    uint32_t *const ring0 = lds + lane * NP;

    const bool has_prev = band > 0 && !(a.dbg & 64);
    Pack<NP> bA[LB][3], bB[LB][3];
    // this band's boundary row as a buffer resource: [x][3 roles][D] int16
    const int row_bytes = W1 * 3 * D * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.bndL + (int64_t)band * W1 * 3 * D), 0, row_bytes, 0x00020000);
    const int voff = lane_off * 2;
    const int px_bytes = 3 * D * 2;
    const int pk = a.xdir > 0 ? px_bytes : -px_bytes;           // byte step per pixel of the sweep order
    const int p0 = a.xdir > 0 ? 0 : (W1 - 1) * px_bytes;
    ChainWait cw;
    if (CHAIN && has_prev) {
        cw.word = a.prog + band - 1;
        cw.err = a.err;
    }
    // FULL = every pixel of the block exists: no guards, so hipcc can count the loads in flight
    // (with a branch between issue and use it falls back to vmcnt(0) and the whole lockstep
    // workgroup waits for HBM latency every block)
    auto lb_t = [&](auto full_c, Pack<NP>(*b)[3], int k0) {
        constexpr bool FULL = decltype(full_c)::value;
        if (CHAIN) cw.step((uint32_t)min(k0 + LB, W1));  // (a block past the row's end waits for nothing new)
#pragma unroll
        for (int u = 0; u < LB; u++) {
            if (FULL || k0 + u < W1) {
                const int so = p0 + (k0 + u) * pk;
#pragma unroll
                for (int d = 0; d < 3; d++) buf_load<NP, LDAUX>(b[u][d], rsrc, voff, so + d * D * 2);
            }
        }
    };
    auto wb_t = [&](auto full_c, Pack<NP>(*b)[3], int off, int u, int k) {  // pixel k = u-th of its block, ring slot off + u
        constexpr bool FULL = decltype(full_c)::value;
        if (!FULL && k > W1) return;
        uint32_t *slot = ring0 + (off + u) * SLOT;
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
        for (int u = 0; u < RING; u++) sweep_write_start_state<NP>(ring0, u, init);
        wg_barrier();
        for (; t < T; t++) wg_barrier();
        return;
    }
    sweep_write_start_state<NP>(ring0, RING - 1, init);
    const std::true_type full{};
    const std::false_type part{};
    // prologue: pixels 0 .. 2*PPS-1, then the rest of block 0
    if (2 * LB <= W1) {
        lb_t(full, bA, 0);
        lb_t(full, bB, LB);
#pragma unroll
        for (int p = 0; p < 2 * PPS; p++) wb_t(full, bA, OFF_A, p, p);
        wg_barrier();
#pragma unroll
        for (int u0 = 2 * PPS; u0 < LB; u0 += PPS) {
#pragma unroll
            for (int p = 0; p < PPS; p++) wb_t(full, bA, OFF_A, u0 + p, u0 + p);
            wg_barrier();
            t++;
        }
    } else {
        lb_t(part, bA, 0);
        lb_t(part, bB, LB);
#pragma unroll
        for (int p = 0; p < 2 * PPS; p++) wb_t(part, bA, OFF_A, p, p);
        wg_barrier();
#pragma unroll
        for (int u0 = 2 * PPS; u0 < LB; u0 += PPS) {
            if (t < T) {
#pragma unroll
                for (int p = 0; p < PPS; p++) wb_t(part, bA, OFF_A, u0 + p, u0 + p);
                wg_barrier();
                t++;
            }
        }
    }
    int k0 = LB;  // bB holds block [k0, k0+LB), bA is free
    // steady state: blocks k0 (in bB), k0+LB (to bA), k0+2LB (to bB) all full -> straight-line
    for (; k0 + 3 * LB <= W1; k0 += 2 * LB) {
        lb_t(full, bA, k0 + LB);
#pragma unroll
        for (int u0 = 0; u0 < LB; u0 += PPS) {
#pragma unroll
            for (int p = 0; p < PPS; p++) wb_t(full, bB, OFF_B, u0 + p, k0 + u0 + p);
            wg_barrier();
        }
        lb_t(full, bB, k0 + 2 * LB);
#pragma unroll
        for (int u0 = 0; u0 < LB; u0 += PPS) {
#pragma unroll
            for (int p = 0; p < PPS; p++) wb_t(full, bA, OFF_A, u0 + p, k0 + LB + u0 + p);
            wg_barrier();
        }
        t += 2 * (LB / PPS);
    }
    // tail: guarded blocks, then idle steps until every row has finished
    for (; t < T; k0 += 2 * LB) {
        lb_t(part, bA, k0 + LB);
#pragma unroll
        for (int u0 = 0; u0 < LB; u0 += PPS) {
            if (t < T) {
#pragma unroll
                for (int p = 0; p < PPS; p++) wb_t(part, bB, OFF_B, u0 + p, k0 + u0 + p);
                wg_barrier();
                t++;
            }
        }
        lb_t(part, bB, k0 + 2 * LB);
#pragma unroll
        for (int u0 = 0; u0 < LB; u0 += PPS) {
            if (t < T) {
#pragma unroll
                for (int p = 0; p < PPS; p++) wb_t(part, bA, OFF_A, u0 + p, k0 + LB + u0 + p);
                wg_barrier();
                t++;
            }
        }
    }
}

// ==== compute wave: one image row ======================================================================
template <int NP, bool PARTIAL, int MODE, bool POSW>
__device__ __forceinline__ void sweep_compute_wave(const Geom &g, const SweepArgs &a, int band, int wave, int lane, uint32_t *lds)
{
    constexpr int LDAUX = SGM_NT_SWEEP_LOADS ? 2 : 0;  // "nt": every byte the sweep loads is used once
    constexpr int PPS = sweep_pps(NP);
    constexpr int RING = sweep_ring(NP);
    constexpr int PB = RING;
    constexpr int SLOT = sweep_slot_dwords(NP);
    constexpr int ROLE = 64 * NP;
    const int R = a.R;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int T = (W1 + PPS - 1) / PPS + 2 * (R - 1);
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const uint32_t init = active ? 0u : SGM_SENT;
    // ring 0 is fed by the loader (row above the band), ring r+1 by compute wave r
    uint32_t *const ring0 = lds + lane * NP;

    const int j = band * R + wave;  // row index in sweep order
    const int y = a.ydir > 0 ? j : H - 1 - j;
    const uint32_t *const prev = ring0 + wave * RING * SLOT;
    uint32_t *const mine = ring0 + (wave + 1) * RING * SLOT;
    // Out-of-image neighbours are ordinary ring slots holding the start state: "pixel -1" lives in
    // slot RING-1 (written before the prologue barrier, not reused until pixel RING-1 exists) and
    // "pixel W1" is written by each producer one step after its last real pixel.  The per-pixel
    // code therefore needs no border branches.
    sweep_write_start_state<NP>(mine, RING - 1, init);
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
    // The reduction leaves the four splats {m, m} in the four rows of one register; its per-lane running maximum
    // (unsigned 32-bit maximum of splats = splat of the maximum) is ONE instruction per pixel -- round 3 took four
    // s_max_u32 on the four v_readlane results: every instruction of any kind costs a wave one issue slot here.
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
    // Full waves (D = 128 NP): a pixel is PXB = 256 NP bytes, a block of PB pixels spans at most 3.8 KB -- inside the 12-bit
    // immediate offset of a buffer instruction.  The pixels of a FULL block are then addressed as ONE scalar offset per
    // block (its lowest address) + an immediate per pixel, instead of one SGPR offset per pixel of the two blocks in flight:
    // round 3's k_sweep_chain held 24 such offsets, ran out of SGPRs (142 spills) and rebuilt them with s_mul_i32 + s_add_i32
    // in the hot loop.  The direction of the sweep is the pass (the engine runs SWEEP_FIRST top-down / left to right and
    // the second pass the other way round: sgm_engine.hip asserts it), so the immediates are compile-time constants.
    constexpr bool IMM = !PARTIAL;
    constexpr int XD = MODE == SWEEP_FIRST ? 1 : -1;
    constexpr int PXB = 256 * NP;
    static_assert((PB - 1) * PXB + 8 < 4096, "a block's pixels must fit the immediate offset field");
    // lowest byte offset of the block of pixels k0 .. k0 + PB - 1 (all of them inside the row), and pixel u's distance from it
    auto block_base = [&](int k0) { return XD > 0 ? k0 * PXB : (W1 - PB - k0) * PXB; };
    auto px_imm = [](int u) { return XD > 0 ? u * PXB : (PB - 1 - u) * PXB; };

    // FULL blocks (all but the last of a row) are straight-line code without guards, so that the
    // scheduler can interleave the independent chains of the PPS pixels of a step
    auto load_block_t = [&](auto full_c, Pack<NP> *cb, Pack<NP> *sb, int k0) {
        constexpr bool FULL = decltype(full_c)::value;
        if constexpr (FULL && IMM) {
            const int sbase = block_base(k0);
#pragma unroll
            for (int u = 0; u < PB; u++) {
                buf_load<NP, LDAUX>(cb[u], Crow, voff + px_imm(u), sbase);
                if (READS_S) buf_load<NP, LDAUX>(sb[u], Srow, voff + px_imm(u), sbase);
            }
            return;
        }
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
    // (vo, so): where this pixel's S goes -- per-lane offset (+ immediate) and scalar offset
    auto pixel = [&](const Pack<NP> &Cp, const Pack<NP> &Sp, const Pack<NP> &QA, const Pack<NP> &QB,
                     const Pack<NP> &QC, int u, int vo, int so) {
        Pack<NP> N0, NA, NB, NC;
        uint32_t r0, rA, rB, rC;
        path_elem<NP, PARTIAL>(Cp, L0, P1s, P2s, active, N0, r0, sr0);
        path_elem<NP, PARTIAL>(Cp, QA, P1s, P2s, active, NA, rA, srA);
        path_elem<NP, PARTIAL>(Cp, QB, P1s, P2s, active, NB, rB, srB);
        path_elem<NP, PARTIAL>(Cp, QC, P1s, P2s, active, NC, rC, srC);
        uint32_t ms[4];  // {m, m} of the directions 0, A, B, C
        uint32_t rows;
        wave_min4_splat(r0, rA, rB, rC, ms, rows);
        hm = max(hm, rows);
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
        if (!PARTIAL || active) buf_store<NP>(Sn, Sst, vo, so);
        return Sn;
    };

    auto compute_block_t = [&](auto full_c, Pack<NP> *cb, Pack<NP> *sb, int k0) {  // k0 % PB == 0
        constexpr bool FULL = decltype(full_c)::value;
        const int sbase = (FULL && IMM) ? block_base(k0) : 0;
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
                    if constexpr (FULL && IMM) Sn[p] = pixel(cb[u], sb[u], QA[p], QB[p], QC[p], u, voff + px_imm(u), sbase);
                    else if (FULL || k0 + u < W1) Sn[p] = pixel(cb[u], sb[u], QA[p], QB[p], QC[p], u, voff, b0 + (k0 + u) * bk);
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
    {
        const uint32_t hmax = wave_max_u32(hm) & 0xffffu;
        if (g.hr && lane == 0) headroom_raise(g.hr + 1, hmax);
    }
    // one step after the last real pixel: the virtual pixel W1 (start state) for the row below
    if (wave < R - 1) {
        sweep_write_start_state<NP>(mine, W1 % RING, init);
        for (int i = 0; i < 2 * (R - 1 - wave); i++) wg_barrier();
    }
}

template <int NP, bool PARTIAL, int MODE, bool POSW>
__global__ __launch_bounds__(SWEEP_MAX_ROWS * 64 + 64) void k_sweep(Geom g, SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int band = blockIdx.x;
    if (wave == a.R) sweep_loader_wave<NP, PARTIAL, false>(g, a, band, lane, lds);
    else sweep_compute_wave<NP, PARTIAL, MODE, POSW>(g, a, band, wave, lane, lds);
}

// ------------------------------------------------------------------------------------------------------
// Chained schedule ("throughput mode", SGM_OPT_SCHEDULE = 2): no boundary pre-pass.  A band takes the
// state of the row above it from the band above, which publishes the path state of its last row to HBM
// while it computes; what a frame then moves per pass is "C read once, S written once" plus that record
// (3 / R of a volume, written and read once) instead of the pre-pass's second and third read of C.
//
// Workgroup = R compute waves + loader wave (wave R) + publisher wave (wave R + 1).  The compute waves are
// those of k_sweep, unchanged.  The publisher reads what compute wave R - 1 leaves in its LDS ring (in
// k_sweep nobody reads that ring) one step later and stores it to the record of band + 1 with write-through
// ("sc1") stores; it then advances the band's progress word to the pixels whose stores are known complete:
// a counted s_waitcnt -- every lockstep step issues the same number of stores (those of pixels that do not
// exist go to an offset past the record and are dropped by the bounds check), so "all but the stores of the
// last KD steps" is a constant -- and never waits for an outstanding store.  The loader of band + 1 polls
// that word (ChainWait) before it prefetches a block.
//
// Order and liveness.  Bands are not tied to blockIdx: a workgroup draws its band from a ticket counter
// when it starts (and again after each band: the grid is persistent, a window of workgroups slides over
// the bands), so the band above any band a workgroup waits for was drawn earlier -- by a workgroup that is
// resident or has finished.  By induction down to band 0, which waits for nothing, every wait ends, for any
// grid size, any dispatch order, and with other kernels sharing the GPU; a workgroup waits only for a lower
// ticket of its own launch.  Barrier counts are those of k_sweep: 1 + T per band for every wave.
//
// What it costs.  A band trails the band above by LAG = 2 (R - 1) + about 17 steps (LDS hand-off to the publisher 1,
// CHAIN_KD 3, the loader's poll one block old 4, a whole block published 4, loaded one block ahead 4, ring lead 1), so
// a single frame's pass is a chain of nbands * LAG + T steps (4K, D = 256, R = 12: about 8800 steps against T = 1814
// for the plain sweep) on T / LAG = about 47 workgroups: slower for one frame, but a frame needs only that many CUs,
// and with several frames in one launch (sgm_pipeline_batch_device) the GPU is full without any pre-pass.
#ifndef SGM_CHAIN_KD
#define SGM_CHAIN_KD 3
#endif
constexpr int CHAIN_KD = SGM_CHAIN_KD;  // a store is taken to be complete when the stores of KD later steps have been issued *and counted*: see chain_publisher_wave

template <int NP, bool PARTIAL>
__device__ __forceinline__ void chain_publisher_wave(const Geom &g, const SweepArgs &a, int band, int lane, uint32_t *lds)
{
    constexpr int PPS = sweep_pps(NP);
    constexpr int RING = sweep_ring(NP);
    constexpr int SLOT = sweep_slot_dwords(NP);
    constexpr int ROLE = 64 * NP;
    constexpr int STORES_PER_STEP = 3 * PPS * (NP == 4 ? 2 : 1) + 1;  // buf_store<4> = two 64-bit stores; + the progress word
    static_assert(CHAIN_KD * STORES_PER_STEP <= 63, "vmcnt is a 6-bit counter");
    const int R = a.R;
    const int W1 = g.W1, D = g.D;
    const int NS = (W1 + PPS - 1) / PPS;
    const int T = NS + 2 * (R - 1);
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const bool publishes = band + 1 < a.nbands;  // the last band of the sweep has nobody below it
    // ring R: written by compute wave R - 1 (its `mine`)
    const uint32_t *const last = lds + lane * NP + R * RING * SLOT;
    const int row_bytes = W1 * 3 * D * 2;
    const __amdgpu_buffer_rsrc_t rec = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.bndL + (int64_t)(band + 1) * W1 * 3 * D), 0, row_bytes, 0x00020000);
    const int voff = lane_off * 2;
    const int px_bytes = 3 * D * 2;
    const int pk = a.xdir > 0 ? px_bytes : -px_bytes;
    const int p0 = a.xdir > 0 ? 0 : (W1 - 1) * px_bytes;
    uint32_t *const word = a.prog + band;
    if (!publishes) {
        for (int n = 0; n <= T; n++) wg_barrier();
        return;
    }
    // interval n = the code between barrier n - 1 and barrier n (interval 0 precedes the prologue barrier,
    // interval T + 1 follows the last barrier).  Compute wave R - 1 finishes its step i (pixels PPS i ..)
    // with barrier 1 + 2 (R - 1) + i, so those pixels are read here in interval n = i + 2 (R - 1) + 2; their
    // ring slots are rewritten four steps later.
    for (int n = 0; n <= T + 1; n++) {
        const int i = n - 2 * (R - 1) - 2;
        if (i >= 0 && i < NS) {
#pragma unroll
            for (int p = 0; p < PPS; p++) {
                const int k = PPS * i + p;
                const uint32_t *s = last + (k % RING) * SLOT;
                Pack<NP> v[3];
#pragma unroll
                for (int d = 0; d < 3; d++) lds_load<NP>(v[d], s + d * ROLE);
                // (a pixel past the row's end: offset past the record, the store is dropped but counted)
                const int so = k < W1 ? p0 + k * pk : row_bytes;
#pragma unroll
                for (int d = 0; d < 3; d++)
                    if (!PARTIAL || active) buf_store<NP, 16>(v[d], rec, voff, so + d * D * 2);
            }
        } else {
            // no pixels in this interval: the same number of (dropped) stores, so that the count below holds
            Pack<NP> z;
            z.fill(0u);
#pragma unroll
            for (int p = 0; p < 3 * PPS; p++)
                if (!PARTIAL || active) buf_store<NP, 16>(z, rec, voff, row_bytes);
        }
        // all stores but those of the last KD intervals (this one included) are complete
        __builtin_amdgcn_s_waitcnt(0x0f70 | ((CHAIN_KD * STORES_PER_STEP) & 15) | (((CHAIN_KD * STORES_PER_STEP) >> 4) << 14));
        const int done = min(max(PPS * (i - CHAIN_KD + 1), 0), W1);
        if (lane == 0) __hip_atomic_store(word, (uint32_t)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n <= T) wg_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the whole row is in HBM
    if (lane == 0) __hip_atomic_store(word, (uint32_t)W1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (The winner-take-all inside the second sweep, SWEEP_LAST, was measured here too -- 12 pairs per launch: 3.25 ms per
// pair against 2.47 + 0.74 for SWEEP_ACCUM + k_wta_t: it saves 2 V of traffic and loses it again to the vector units --
// and is not instantiated.)
constexpr int CHAIN_MAX_ROWS = 12;  // 12 compute waves + loader + publisher = 14 waves: three compute waves on every SIMD
template <int NP, bool PARTIAL, int MODE>
__global__ __launch_bounds__(CHAIN_MAX_ROWS * 64 + 128) void k_sweep_chain(Geom g, SweepArgs a, ChainFrames fr)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    uint32_t *const ticket = lds + (a.R + 1) * sweep_ring(NP) * sweep_slot_dwords(NP);  // one word behind the rings
    for (;;) {
        if (threadIdx.x == 0) *ticket = atomicAdd(a.ctl, 1u);
        wg_barrier();
        const int tk = __builtin_amdgcn_readfirstlane((int)*ticket);
        // (the next write of the ticket word comes after the 1 + T barriers of the band: every wave has read it by then)
        if (tk >= a.nbands * fr.nf) break;
        const int f = tk % fr.nf, band = tk / fr.nf;
        a.C = fr.C[f];
        a.S = fr.S[f];
        a.bndL = fr.bnd[f];
        a.prog = a.ctl + 1 + f * a.nbands;
        g.hr = fr.hr[f];
        if (wave == a.R) sweep_loader_wave<NP, PARTIAL, true>(g, a, band, lane, lds);
        else if (wave == a.R + 1) chain_publisher_wave<NP, PARTIAL>(g, a, band, lane, lds);
        else sweep_compute_wave<NP, PARTIAL, MODE, true>(g, a, band, wave, lane, lds);
    }
}

}  // namespace sgm

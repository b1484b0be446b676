// kernels_sweep.h -- four SGM directions fused in one sweep over the image.
//
// Replaces one whole pass of upstream computeDisparitySGBM (/root/reference/main.ipynb:668;
// SURVEY.md A.5): for every pixel the path from the previous pixel of the same row and the
// three paths from the previously processed row, S = sat(sum), and in the last pass the
// winner-take-all scan.  Same arithmetic as k_path (path_recur / wta_pixel), different
// schedule: traffic per pass is "read C once, write S once" instead of 2-3 volumes per direction.
//
// Schedule.  The image is cut into bands of R rows (in sweep order).  One workgroup owns one
// band: R compute waves (one image row each) + 1 loader wave.  The compute waves run the row
// recurrence in lockstep, wave r two pixels behind wave r-1, one workgroup barrier per pixel
// step: the state a row needs from the row above (three L vectors + minima per pixel) travels
// through a 4-pixel LDS ring per wave and never touches HBM.  Bands do not wait for each other:
// the state of the row above a band comes from the PATH_BOUNDARY pre-pass of k_path (three
// read-only line scans that store L only at band boundaries); the loader wave streams it from
// HBM into a 16-pixel LDS ring, one prefetch block ahead of wave 0.
#pragma once
#include "kernels_path.h"

namespace sgm {

enum { SWEEP_FIRST = 0, SWEEP_ACCUM = 1, SWEEP_LAST = 2 };

struct SweepArgs {
    int ydir, xdir;  // (+1,+1): rows top->bottom, x ascending; (-1,-1): second pass of MODE_HH
    int R;           // rows per band = compute waves per workgroup
    const int16_t *C;
    int16_t *S;
    const int16_t *bndL;  // [band][x][3][D]
    const int32_t *bndM;  // [band][x][4]
    uint2 *wta;
    int keepS;
};

constexpr int SWEEP_RING = 4;    // pixels of hand-off state kept per compute wave
constexpr int SWEEP_BRING = 16;  // pixels of boundary state kept ahead of wave 0
constexpr int SWEEP_MAX_ROWS = 9; // compute waves per workgroup (640 threads -> 168 VGPRs per lane)

__host__ __device__ constexpr int sweep_slot_dwords(int NP) { return 192 * NP + 4; }
static inline size_t sweep_lds_bytes(int NP, int R)
{
    return (size_t)(R * SWEEP_RING + SWEEP_BRING) * sweep_slot_dwords(NP) * 4;
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

template <int NP, bool PARTIAL, int MODE>
__global__ __launch_bounds__(SWEEP_MAX_ROWS * 64 + 64) void k_sweep(Geom g, SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int PB = NP == 4 ? 4 : 8;  // prefetch block (pixels); smaller for wide lanes to stay in registers
    constexpr int SLOT = sweep_slot_dwords(NP);
    constexpr int MOFF = 192 * NP;  // minima inside a slot
    const int R = a.R;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int band = blockIdx.x;
    const int W1 = g.W1, D = g.D, H = g.H;
    const int T = W1 + 2 * (R - 1);  // lockstep steps
    const bool active = !PARTIAL || (2 * NP * lane < D);
    const int lane_off = active ? 2 * NP * lane : 0;
    const uint32_t init = active ? 0u : SGM_SENT;
    uint32_t *const rings = lds;                              // [R][SWEEP_RING][SLOT]
    uint32_t *const bring = lds + R * SWEEP_RING * SLOT;      // [SWEEP_BRING][SLOT]

    if (wave == R) {
        // ================= loader wave: boundary state HBM -> LDS, PB pixels ahead ===============
        const bool has_prev = band > 0;
        Pack<NP> bA[PB][3], bB[PB][3];
        uint4 mA = make_uint4(0, 0, 0, 0), mB = make_uint4(0, 0, 0, 0);
        auto xof = [&](int k) { return a.xdir > 0 ? k : W1 - 1 - k; };
        auto lb = [&](Pack<NP>(*b)[3], uint4 &mm, int k0) {
            if (!has_prev) return;
#pragma unroll
            for (int u = 0; u < PB; u++) {
                if (k0 + u < W1) {
                    const int64_t px = (int64_t)band * W1 + xof(k0 + u);
#pragma unroll
                    for (int d = 0; d < 3; d++) b[u][d].load(a.bndL + (px * 3 + d) * D + lane_off);
                }
            }
            if (lane < PB && k0 + lane < W1)
                mm = *reinterpret_cast<const uint4 *>(a.bndM + ((int64_t)band * W1 + xof(k0 + lane)) * 4);
        };
        auto wb = [&](Pack<NP>(*b)[3], const uint4 &mm, int u, int k) {
            uint32_t *slot = bring + (k & (SWEEP_BRING - 1)) * SLOT;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                Pack<NP> v;
                if (has_prev) {
                    v = b[u][d];
                    if (PARTIAL && !active) v.fill(SGM_SENT);
                } else {
                    v.fill(init);
                }
                lds_store<NP>(v, slot + d * 64 * NP + lane * NP);
            }
            const uint32_t m0 = has_prev ? __builtin_amdgcn_readlane(mm.x, u) : 0u;
            const uint32_t m1 = has_prev ? __builtin_amdgcn_readlane(mm.y, u) : 0u;
            const uint32_t m2 = has_prev ? __builtin_amdgcn_readlane(mm.z, u) : 0u;
            if (lane == 0) {
                slot[MOFF + 0] = m0;
                slot[MOFF + 1] = m1;
                slot[MOFF + 2] = m2;
            }
        };
        lb(bA, mA, 0);
        lb(bB, mB, PB);
#pragma unroll
        for (int u = 0; u < PB; u++)
            if (u < W1) wb(bA, mA, u, u);
        wg_barrier();  // prologue barrier
        int t = 0;
        for (int k0 = PB; t < T; k0 += 2 * PB) {
            lb(bA, mA, k0 + PB);
#pragma unroll
            for (int u = 0; u < PB; u++) {
                if (t < T) {
                    if (k0 + u < W1) wb(bB, mB, u, k0 + u);
                    wg_barrier();
                    t++;
                }
            }
            lb(bB, mB, k0 + 2 * PB);
#pragma unroll
            for (int u = 0; u < PB; u++) {
                if (t < T) {
                    if (k0 + PB + u < W1) wb(bA, mA, u, k0 + PB + u);
                    wg_barrier();
                    t++;
                }
            }
        }
        return;
    }

    // ================= compute wave: one image row =============================================
    const int j = band * R + wave;  // row index in sweep order
    const int y = a.ydir > 0 ? j : H - 1 - j;
    wg_barrier();  // prologue barrier
    if (j >= H) {  // row past the image (last band): keep the barrier count, do nothing
        for (int t = 0; t < T; t++) wg_barrier();
        return;
    }
    const uint32_t *prev = wave == 0 ? bring : rings + (wave - 1) * SWEEP_RING * SLOT;
    const int pmask = wave == 0 ? SWEEP_BRING - 1 : SWEEP_RING - 1;
    uint32_t *const mine = rings + wave * SWEEP_RING * SLOT;
    const uint32_t P1s = splat16((uint32_t)g.P1);
    const uint32_t P2 = (uint32_t)g.P2;
    constexpr bool READS_S = MODE != SWEEP_FIRST;

    for (int i = 0; i < 2 * wave; i++) wg_barrier();  // start two pixels behind the row above

    Pack<NP> L0;
    L0.fill(init);
    uint32_t m0 = 0;
    Pack<NP> cA[PB], cB[PB], sA[PB], sB[PB];
    const int64_t rowbase = (int64_t)y * W1;
    auto xof = [&](int k) { return a.xdir > 0 ? k : W1 - 1 - k; };

    auto load_block = [&](Pack<NP> *cb, Pack<NP> *sb, int k0) {
#pragma unroll
        for (int u = 0; u < PB; u++) {
            if (k0 + u < W1) {
                const int64_t off = (rowbase + xof(k0 + u)) * D + lane_off;
                cb[u].load(a.C + off);
                if (READS_S) sb[u].load(a.S + off);
            }
        }
    };

    auto compute_block = [&](Pack<NP> *cb, Pack<NP> *sb, int k0) {
#pragma unroll
        for (int u = 0; u < PB; u++) {
            const int k = k0 + u;
            if (k < W1) {
                const int x = xof(k);
                const int km = a.xdir > 0 ? k - 1 : k + 1;  // step index of pixel x-1
                const int kp = a.xdir > 0 ? k + 1 : k - 1;  // step index of pixel x+1
                const bool hm = x > 0, hp = x < W1 - 1;
                // ---- state of the row above (LDS) ----
                Pack<NP> Q1, Q2, Q3;
                uint32_t q1 = 0, q2, q3 = 0;
                Q1.fill(init);
                Q3.fill(init);
                if (hm) {
                    const uint32_t *s = prev + (km & pmask) * SLOT;
                    lds_load<NP>(Q1, s + 0 * 64 * NP + lane * NP);
                    q1 = s[MOFF + 0];
                }
                {
                    const uint32_t *s = prev + (k & pmask) * SLOT;
                    lds_load<NP>(Q2, s + 1 * 64 * NP + lane * NP);
                    q2 = s[MOFF + 1];
                }
                if (hp) {
                    const uint32_t *s = prev + (kp & pmask) * SLOT;
                    lds_load<NP>(Q3, s + 2 * 64 * NP + lane * NP);
                    q3 = s[MOFF + 2];
                }
                // ---- four recurrences ----
                Pack<NP> N0, N1, N2, N3;
                uint32_t n0, n1, n2, n3;
                path_recur<NP, PARTIAL>(cb[u], L0, m0, P1s, P2, active, N0, n0);
                path_recur<NP, PARTIAL>(cb[u], Q1, q1, P1s, P2, active, N1, n1);
                path_recur<NP, PARTIAL>(cb[u], Q2, q2, P1s, P2, active, N2, n2);
                path_recur<NP, PARTIAL>(cb[u], Q3, q3, P1s, P2, active, N3, n3);
                // ---- hand the three vertical states to the row below ----
                {
                    uint32_t *s = mine + (k & (SWEEP_RING - 1)) * SLOT;
                    lds_store<NP>(N1, s + 0 * 64 * NP + lane * NP);
                    lds_store<NP>(N2, s + 1 * 64 * NP + lane * NP);
                    lds_store<NP>(N3, s + 2 * 64 * NP + lane * NP);
                    if (lane == 0) {
                        s[MOFF + 0] = n1;
                        s[MOFF + 1] = n2;
                        s[MOFF + 2] = n3;
                    }
                }
                // ---- S ----
                const int64_t off = (rowbase + x) * D + lane_off;
                Pack<NP> Sn;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    uint32_t s = pk_adds_s(pk_adds_s(N0.r[i], N1.r[i]), pk_adds_s(N2.r[i], N3.r[i]));
                    if (READS_S) s = pk_adds_s(s, sb[u].r[i]);
                    Sn.r[i] = s;
                }
                if (MODE != SWEEP_LAST || a.keepS) {
                    if (active) Sn.store(a.S + off);
                }
                if (MODE == SWEEP_LAST) {
                    const uint2 rec = wta_pixel<NP, PARTIAL>(Sn, lane, active, D, g.uniq);
                    if (lane == 0) a.wta[(int64_t)y * g.W + x + g.minX1] = rec;
                }
                L0 = N0;
                m0 = n0;
                wg_barrier();
            }
        }
    };

    load_block(cA, sA, 0);
    for (int k0 = 0; k0 < W1; k0 += 2 * PB) {
        load_block(cB, sB, k0 + PB);
        compute_block(cA, sA, k0);
        load_block(cA, sA, k0 + 2 * PB);
        compute_block(cB, sB, k0 + PB);
    }
    for (int i = 0; i < 2 * (R - 1 - wave); i++) wg_barrier();
}

}  // namespace sgm

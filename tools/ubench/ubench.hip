// Microbenchmarks of the instruction patterns of the SGM recurrence on gfx950 (timing only).
// Build: hipcc -O3 --offload-arch=gfx950 -I../../stereo_reconstruction_cv_amd/csrc -o ubench ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_path.h"
using namespace sgm;

template <int PAT>
__global__ void k(uint32_t *out, unsigned long long *cyc, int iters, uint32_t seed)
{
    const int lane = threadIdx.x & 63;
    Pack<2> L, C, Q1, Q2, Q3;
    L.r[0] = seed * (lane + 1) & 0x0fff0fff; L.r[1] = (seed + lane) & 0x0fff0fff;
    C.r[0] = (seed ^ lane) & 0x03ff03ff; C.r[1] = (seed + 3 * lane) & 0x03ff03ff;
    Q1 = L; Q2 = C; Q3 = L;
    const uint32_t P1s = splat16(100), P2s = splat16(3000);
    uint32_t acc = 0;
    ShiftRegs s0, s1, s2, s3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (PAT == 0) {  // 16 dependent packed mins
#pragma unroll
            for (int i = 0; i < 16; i++) L.r[0] = pk_min_s(pk_add(L.r[0], P1s), C.r[i & 1]);
        } else if (PAT == 1) {  // reduce + normalise round trip (loop carried)
            uint32_t m = wave_min_pk(pk_min_s(L.r[0], L.r[1]));
            Pack<2> N;
            path_normalise<2, false>(L, halves_min(m), true, N);
            L.r[0] = pk_add(N.r[0], C.r[0]); L.r[1] = pk_add(N.r[1], C.r[1]);
        } else if (PAT == 2) {  // one direction element-wise only (no reduce)
            Pack<2> N; uint32_t r;
            path_elem<2, false>(C, L, P1s, P2s, true, N, r, s0);
            L = N; acc ^= r;
        } else if (PAT == 3) {  // one full direction step (k_path body)
            Pack<2> N, Nn; uint32_t r;
            path_elem<2, false>(C, L, P1s, P2s, true, N, r, s0);
            path_normalise<2, false>(N, halves_min(wave_min_pk(r)), true, Nn);
            L = Nn;
        } else if (PAT == 4) {  // sweep pixel: 4 directions, 2 batched reductions
            Pack<2> N0, NA, NB, NC; uint32_t r0, rA, rB, rC;
            path_elem<2, false>(C, L, P1s, P2s, true, N0, r0, s0);
            path_elem<2, false>(C, Q1, P1s, P2s, true, NA, rA, s1);
            path_elem<2, false>(C, Q2, P1s, P2s, true, NB, rB, s2);
            path_elem<2, false>(C, Q3, P1s, P2s, true, NC, rC, s3);
            const uint32_t m0A = wave_min_pk(pk_min_s(pack_lo(r0, rA), pack_hi(r0, rA)));
            const uint32_t mBC = wave_min_pk(pk_min_s(pack_lo(rB, rC), pack_hi(rB, rC)));
            path_normalise<2, false>(N0, m0A & 0xffffu, true, L);
            path_normalise<2, false>(NA, m0A >> 16, true, Q1);
            path_normalise<2, false>(NB, mBC & 0xffffu, true, Q2);
            path_normalise<2, false>(NC, mBC >> 16, true, Q3);
        } else if (PAT == 5) {  // wave_shr / wave_shl DPP pairs only
#pragma unroll
            for (int i = 0; i < 8; i++) {
                L.r[0] = pk_min_s(from_lower_lane(L.r[1], SGM_SENT), C.r[0]);
                L.r[1] = pk_min_s(from_upper_lane(L.r[0], SGM_SENT), C.r[1]);
            }
        } else if (PAT == 6) {  // 12 row-local DPP steps (quad/mirror), dependent
#pragma unroll
            for (int i = 0; i < 3; i++) {
                L.r[0] = pk_min_s(pk_add(L.r[0], P1s), dpp_view<DPP_QUAD_1032>(L.r[0]));
                L.r[0] = pk_min_s(pk_add(L.r[0], P1s), dpp_view<DPP_QUAD_2301>(L.r[0]));
                L.r[0] = pk_min_s(pk_add(L.r[0], P1s), dpp_view<DPP_ROW_HALF_MIRROR>(L.r[0]));
                L.r[0] = pk_min_s(pk_add(L.r[0], P1s), dpp_view<DPP_ROW_MIRROR>(L.r[0]));
            }
        } else if (PAT == 7) {  // readlane -> SALU -> VALU round trip x4
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t m = __builtin_amdgcn_readlane(L.r[0], 63);
                L.r[0] = pk_sub(pk_add(L.r[0], C.r[0]), splat16(halves_min(m)));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = L.r[0] ^ L.r[1] ^ acc ^ Q1.r[0] ^ Q2.r[1] ^ Q3.r[0];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int PAT> void run(const char *name, int instrs_hint)
{
    const int iters = 4000;
    for (int wpb : {1, 4, 8, 12, 16}) {  // waves per block = per CU (1 block per CU)
        uint32_t *out; unsigned long long *cyc;
        hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
        hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters, 12345u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * wpb);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        // s_memtime counts at 100 MHz constant clock? report both ticks and wall
        printf("%-28s waves/CU %2d (%.2f/SIMD): %8.1f memtime-ticks/iter  wall %7.1f ns/iter/wave-slot  (%d instr hint)\n", name, wpb, wpb / 4.0,
               s / h.size() / iters, ms * 1e6 / iters, instrs_hint);
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<0>("16 dep pk_add+pk_min", 32);
    run<6>("12 dep row-DPP+add+min", 36);
    run<5>("16 wave_shr/shl + min", 48);
    run<7>("4x readlane->SALU->VALU", 24);
    run<1>("reduce+normalise", 24);
    run<2>("elem 1 dir", 19);
    run<3>("k_path step (no mem)", 40);
    run<4>("sweep pixel (no mem/LDS)", 125);
    return 0;
}

// Probe for DESIGN.md 4.6: the two ways k_pix has put its row address together.  hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only:
// k_bad (the intermediate form of round 3: the int that __builtin_amdgcn_readfirstlane returns is sign-extended over the high half)
// shows "s_bfe_i64 ..., 0x200000" in front of the s_or_b64; k_good (the committed form) has neither.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v2u32 *crec_ptr;
// suspected intermediate form: the halves reassembled without going through uint32_t
__global__ void k_bad(const uint2 *lrec, int64_t off, uint32_t *out)
{
    const uint64_t la = (uint64_t)(uintptr_t)(lrec + off);
    const crec_ptr lrow = (crec_ptr)(uintptr_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(la >> 32)) << 32) |
                                              __builtin_amdgcn_readfirstlane((uint32_t)la));
    v2u32 r = lrow[0];
    out[threadIdx.x] = r.x + r.y;
}
// committed form
__global__ void k_good(const uint2 *lrec, int64_t off, uint32_t *out)
{
    const uint64_t la = (uint64_t)(uintptr_t)(lrec + off);
    const uint32_t la_lo = __builtin_amdgcn_readfirstlane((uint32_t)la), la_hi = __builtin_amdgcn_readfirstlane((uint32_t)(la >> 32));
    const crec_ptr lrow = (crec_ptr)(uintptr_t)(((uint64_t)la_hi << 32) | la_lo);
    v2u32 r = lrow[0];
    out[threadIdx.x] = r.x + r.y;
}

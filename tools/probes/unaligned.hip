// Probe: are byte-aligned 32-bit loads from global memory and from LDS single instructions on gfx950 under ROCm's default
// (unaligned access mode), and do they return the right bytes?  hipcc -O3 --offload-arch=gfx950 -o unaligned unaligned.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
struct __attribute__((packed)) U32 { uint32_t v; };
__global__ void k_global(const uint8_t *p, int off, uint32_t *out)
{
    const U32 *q = reinterpret_cast<const U32 *>(p + off + 4 * threadIdx.x);
    out[threadIdx.x] = q->v;
}
__global__ void k_lds(const uint8_t *p, int off, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = p[i];
    __syncthreads();
    const U32 *q = reinterpret_cast<const U32 *>(s + off + 4 * threadIdx.x);
    out[threadIdx.x] = q->v;
}
int main()
{
    std::vector<uint8_t> h(2048);
    for (int i = 0; i < 2048; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, 2048); hipMalloc(&o, 256);
    hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
    int bad = 0;
    for (int which = 0; which < 2; which++)
        for (int off = 0; off < 8; off++) {
            if (which == 0) hipLaunchKernelGGL(k_global, dim3(1), dim3(64), 0, 0, d, off, o);
            else hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, d, off, o);
            uint32_t r[64];
            hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; l++) {
                uint32_t w; memcpy(&w, &h[off + 4 * l], 4);
                if (w != r[l]) bad++;
            }
        }
    printf("unaligned probe: %d mismatches\n", bad);
    return bad != 0;
}

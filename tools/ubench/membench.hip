// Access-pattern microbenchmark: one wave per column walking down rows of a [H][W1][256] int16 volume
// (512 B per step), prefetching PF steps ahead in registers; variants of the row/tile layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int PF>
__global__ __launch_bounds__(64) void walk(const uint2 *__restrict__ v, uint32_t *out, int H, int W1, int TR /*rows per tile*/)
{
    const int lane = threadIdx.x, x = blockIdx.x;
    // tiled layout: [y / TR][x][y % TR][64 lanes] of uint2 ; TR = 1 is the plain row-major volume
    auto addr = [&](int y) { return (((int64_t)(y / TR) * W1 + x) * TR + (y % TR)) * 64 + lane; };
    uint2 buf[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) buf[i] = v[addr(i)];
    uint32_t acc = 0;
    for (int y = 0; y < H; y += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            uint2 c = buf[i];
            if (y + PF + i < H) buf[i] = v[addr(y + PF + i)];
            acc += c.x ^ c.y;
        }
    }
    out[x * 64 + lane] = acc;
}
int main()
{
    const int H = 2160, W1 = 3584;
    const size_t n = (size_t)H * W1 * 64;
    uint2 *v; uint32_t *out;
    hipMalloc(&v, n * 8); hipMalloc(&out, W1 * 64 * 4);
    hipMemset(v, 1, n * 8);
    for (int TR : {1, 8, 16, 72}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(walk<16>, dim3(W1), dim3(64), 0, 0, v, out, (H / TR) * TR, W1, TR);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("column walk, tile rows %2d, prefetch 16: %.3f ms  %.2f TB/s\n", TR, ms, n * 8 / ms / 1e9);
        }
    }
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(walk<32>, dim3(W1), dim3(64), 0, 0, v, out, H, W1, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("column walk, tile rows  1, prefetch 32: %.3f ms  %.2f TB/s\n", ms, n * 8 / ms / 1e9);
    }
    return 0;
}

// Hazard probe (gfx950): does a SALU write of the SGPR that a just-issued MUBUF store uses as its
// soffset corrupt the store?  Variant 0: s_add right after the store; variant 1: s_nop 7 between.
// Build: hipcc -O2 --offload-arch=gfx950 -o soffset_war soffset_war.hip ; run: ./soffset_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(640) void probe(unsigned *out, int iters, size_t bytes_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char *base = (char *)out + (size_t)wave * bytes_per_wave;
    v4u rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32) & 0xffffu);
    rsrc.z = __builtin_amdgcn_readfirstlane((unsigned)bytes_per_wave);
    rsrc.w = 0x00020000u;
    int voff = lane * 16;
    int soff = 0;
    for (int i = 0; i < iters; i++) {
        v4u d;
        d.x = (wave << 16) | (i << 6) | lane;
        d.y = d.x ^ 0x11111111u;
        d.z = d.x ^ 0x22222222u;
        d.w = d.x ^ 0x33333333u;
        int junk;
        if (VARIANT == 0)
            asm volatile("buffer_store_dwordx4 %1, %2, %3, %4 offen\n\ts_add_i32 %0, %4, 0x7f0" : "=s"(junk) : "v"(d), "v"(voff), "s"(rsrc), "0"(soff) : "memory");
        else
            asm volatile("buffer_store_dwordx4 %1, %2, %3, %4 offen\n\ts_nop 7\n\ts_add_i32 %0, %4, 0x7f0" : "=s"(junk) : "v"(d), "v"(voff), "s"(rsrc), "0"(soff) : "memory");
        soff = (junk - 0x7f0) + 1024;
    }
}

template <int VARIANT> int run(int blocks, int iters)
{
    const int waves = blocks * 10;
    const size_t bpw = (size_t)iters * 1024;
    unsigned *d;
    hipMalloc(&d, waves * bpw);
    hipMemset(d, 0xff, waves * bpw);
    hipLaunchKernelGGL(probe<VARIANT>, dim3(blocks), dim3(640), 0, 0, d, iters, bpw);
    hipDeviceSynchronize();
    std::vector<unsigned> h(waves * bpw / 4);
    hipMemcpy(h.data(), d, waves * bpw, hipMemcpyDeviceToHost);
    long bad = 0;
    int shown = 0;
    for (int w = 0; w < waves; w++)
        for (int i = 0; i < iters; i++)
            for (int l = 0; l < 64; l++) {
                const unsigned x = ((unsigned)w << 16) | (i << 6) | l;
                const unsigned *p = &h[((size_t)w * bpw + (size_t)i * 1024 + l * 16) / 4];
                const unsigned e[4] = {x, x ^ 0x11111111u, x ^ 0x22222222u, x ^ 0x33333333u};
                for (int k = 0; k < 4; k++)
                    if (p[k] != e[k]) {
                        bad++;
                        if (shown++ < 12) printf("  wave %d iter %d lane %d dword %d: got %08x want %08x\n", w, i, l, k, p[k], e[k]);
                    }
            }
    printf("variant %d: %ld bad dwords of %zu\n", VARIANT, bad, h.size());
    hipFree(d);
    return bad != 0;
}

int main()
{
    int r = 0;
    for (int rep = 0; rep < 3; rep++) {
        r |= run<0>(512, 256);
        r |= run<1>(512, 256) << 1;
    }
    return 0;
}

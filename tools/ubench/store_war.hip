// Hazard probe (gfx950): a buffer_store_dwordx4 is followed, ~100 VALU instructions later, by an
// EXEC-masked VALU write to its first data register.  Does the stored data stay intact?
//   variant 0: dwordx4 store + masked write     variant 1: dwordx4 store, no masked write
//   variant 2: two dwordx2 stores + masked write
// Each wave also streams loads so that the memory pipeline is busy.
// Build: hipcc -O2 --offload-arch=gfx950 -o store_war store_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(640) void probe(unsigned *out, const unsigned *junk_in, unsigned *sink, unsigned *sink2, int iters,
                                             size_t bytes_per_wave, size_t junk_words)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char *base = (char *)out + (size_t)wave * bytes_per_wave;
    v4u rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32) & 0xffffu);
    rsrc.z = __builtin_amdgcn_readfirstlane((unsigned)bytes_per_wave);
    rsrc.w = 0x00020000u;
    const int voff = lane * 16;
    const size_t s2a = (size_t)(sink2 + (size_t)wave * 2);
    const unsigned long long s2p = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(s2a >> 32)) << 32) |
                                   __builtin_amdgcn_readfirstlane((unsigned)s2a);
    unsigned acc = 0;
    size_t jpos = ((size_t)wave * 7919u * 64u) % junk_words;
    for (int i = 0; i < iters; i++) {
        // background loads (kept live through acc)
        const uint4 j0 = *reinterpret_cast<const uint4 *>(junk_in + ((jpos + lane * 4) % junk_words));
        const uint4 j1 = *reinterpret_cast<const uint4 *>(junk_in + ((jpos + 4096 + lane * 4) % junk_words));
        jpos = (jpos + 8192 + 64 * 13) % junk_words;
        v4u d;
        d.x = (wave << 16) | (i << 6) | lane;
        d.y = d.x ^ 0x11111111u;
        d.z = d.x ^ 0x22222222u;
        d.w = d.x ^ 0x33333333u;
        const int soff = i * 1024;
        unsigned t = lane * 3 + i;
        // store, ~100 dependent VALU ops on t, then a masked write of the first data register
        asm volatile(
            "v_mov_b32 v120, %1\n\t"
            "v_mov_b32 v121, %2\n\t"
            "v_mov_b32 v122, %3\n\t"
            "v_mov_b32 v123, %4\n\t"
            "s_nop 4\n\t"
            ".if %9 == 2\n\t"
            "buffer_store_dwordx2 v[120:121], %5, %6, %7 offen\n\t"
            "buffer_store_dwordx2 v[122:123], %5, %6, %7 offen offset:8\n\t"
            ".else\n\t"
            "buffer_store_dwordx4 v[120:123], %5, %6, %7 offen\n\t"
            ".endif\n\t"
            ".rept 100\n\t"
            "v_add_u32 %0, %0, %8\n\t"
            ".endr\n\t"
            ".if %9 != 1\n\t"
            "s_mov_b64 s[10:11], exec\n\t"
            "s_mov_b64 exec, 0x20\n\t"       // lane 5 only
            "v_mov_b32 v120, 0xdeadbeef\n\t"
            "global_store_short %10, v120, %11 offset:6\n\t"
            "s_mov_b64 exec, 1\n\t"       // lane 0 only
            "v_mov_b32 v120, 0x12345678\n\t"
            "global_store_dword %10, v120, %11\n\t"
            "s_mov_b64 exec, s[10:11]\n\t"
            ".endif\n\t"
            ".rept 30\n\t"
            "v_add_u32 %0, %0, %8\n\t"
            ".endr\n\t"
            : "+v"(t) : "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w), "v"(voff), "s"(rsrc), "s"(soff), "v"(lane), "n"(VARIANT), "v"(0), "s"(s2p)
            : "memory", "s10", "s11", "v120", "v121", "v122", "v123");
        acc += t + j0.x + j0.w + j1.y + d.x;
    }
    sink[wave * 64 + lane] = acc;
}

template <int VARIANT> int run(int blocks, int iters, const unsigned *junk, size_t junk_words)
{
    const int waves = blocks * 10;
    const size_t bpw = (size_t)iters * 1024;
    unsigned *d, *sink, *sink2;
    hipMalloc(&sink2, waves * 8 + 64);
    hipMalloc(&d, waves * bpw);
    hipMalloc(&sink, waves * 64 * 4);
    hipMemset(d, 0xff, waves * bpw);
    hipLaunchKernelGGL(probe<VARIANT>, dim3(blocks), dim3(640), 0, 0, d, junk, sink, sink2, iters, bpw, junk_words);
    hipDeviceSynchronize();
    std::vector<unsigned> h(waves * bpw / 4);
    hipMemcpy(h.data(), d, waves * bpw, hipMemcpyDeviceToHost);
    long bad = 0;
    int shown = 0;
    long lanehist[64] = {0}, dwhist[4] = {0};
    for (int w = 0; w < waves; w++)
        for (int i = 0; i < iters; i++)
            for (int l = 0; l < 64; l++) {
                const unsigned x = ((unsigned)w << 16) | (i << 6) | l;
                const unsigned *p = &h[((size_t)w * bpw + (size_t)i * 1024 + l * 16) / 4];
                const unsigned e[4] = {x, x ^ 0x11111111u, x ^ 0x22222222u, x ^ 0x33333333u};
                for (int k = 0; k < 4; k++)
                    if (p[k] != e[k]) {
                        bad++;
                        lanehist[l]++;
                        dwhist[k]++;
                        if (shown++ < 6) printf("  wave %d iter %d lane %d dword %d: got %08x want %08x\n", w, i, l, k, p[k], e[k]);
                    }
            }
    printf("variant %d: %ld bad dwords of %zu; by dword %ld %ld %ld %ld; lanes:", VARIANT, bad, h.size(), dwhist[0], dwhist[1], dwhist[2], dwhist[3]);
    for (int l = 0; l < 64; l++) if (lanehist[l]) printf(" %d:%ld", l, lanehist[l]);
    printf("\n");
    hipFree(d);
    hipFree(sink);
    return bad != 0;
}

int main()
{
    const size_t junk_words = (size_t)64 << 20;  // 256 MB
    unsigned *junk;
    hipMalloc(&junk, junk_words * 4);
    hipMemset(junk, 1, junk_words * 4);
    for (int rep = 0; rep < 2; rep++) {
        run<0>(512, 128, junk, junk_words);
        run<1>(512, 128, junk, junk_words);
        run<2>(512, 128, junk, junk_words);
    }
    return 0;
}

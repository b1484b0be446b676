// sgm_device.h -- wave64 / packed-int16 building blocks for gfx950 (CDNA4).
//
// Layout convention used by every cost-volume kernel: a volume is int16 [y][xi][d] with d
// fastest.  One wavefront owns one pixel's D disparities: lane l holds NP packed pairs
// (32-bit registers of two int16), pair i of lane l = disparities d = 2*(NP*l + i) + {0,1}.
// NP = 1, 2, 4 covers D <= 128, 256, 512; lanes whose first disparity is >= D are idle
// ("partial" waves, D % 16 == 0 makes a lane either fully valid or fully idle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgm {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bits(v2s v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t bits(v2u v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ v2s as_s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ v2u as_u(uint32_t v) { return __builtin_bit_cast(v2u, v); }
__device__ __forceinline__ uint32_t splat16(uint32_t v) { return (v & 0xffffu) | (v << 16); }

// packed int16 arithmetic on raw 32-bit registers
__device__ __forceinline__ uint32_t pk_min_s(uint32_t a, uint32_t b) { return bits(__builtin_elementwise_min(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return bits(as_u(a) + as_u(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return bits(as_u(a) - as_u(b)); }
__device__ __forceinline__ uint32_t pk_adds_s(uint32_t a, uint32_t b) { return bits(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }  // v_pk_add_i16 clamp
__device__ __forceinline__ uint32_t pk_subs_u(uint32_t a, uint32_t b) { return bits(__builtin_elementwise_sub_sat(as_u(a), as_u(b))); }  // max(0, a-b)
__device__ __forceinline__ uint32_t pk_max_u(uint32_t a, uint32_t b) { return bits(__builtin_elementwise_max(as_u(a), as_u(b))); }
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { return bits(__builtin_elementwise_min(as_u(a), as_u(b))); }
__device__ __forceinline__ uint32_t pk_shr_u(uint32_t a, int s) { return bits(as_u(a) >> (unsigned short)s); }

// DPP controls (GFX9 encodings)
enum : int {
    DPP_QUAD_1032 = 0xB1,
    DPP_QUAD_2301 = 0x4E,
    DPP_ROW_HALF_MIRROR = 0x141,
    DPP_ROW_MIRROR = 0x140,
    DPP_WAVE_SHL1 = 0x130,  // lane i reads lane i+1
    DPP_WAVE_SHR1 = 0x138,  // lane i reads lane i-1
};

// lane i <- lane i-1 ; lane 0 <- fill
__device__ __forceinline__ uint32_t from_lower_lane(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}
// lane i <- lane i+1 ; lane 63 <- fill
__device__ __forceinline__ uint32_t from_upper_lane(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, DPP_WAVE_SHL1, 0xf, 0xf, false);
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One butterfly / broadcast step of a wave-wide reduction: y = x as seen through a DPP control.
// mov_dpp (no "old" operand) lets the compiler emit a single v_mov_b32_dpp without a copy.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_view(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, ROW_MASK, 0xf, true);
}
enum : int { DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143 };

// min over the 64 lanes of both int16 halves of x at once; the wave-uniform result packs
// {min of low halves, min of high halves}.  4 butterfly steps inside each row of 16 lanes, then
// row_bcast 15 / 31 fold the rows into lane 63 (rows masked off in those two steps hold
// garbage afterwards -- only lane 63 is read).
__device__ __forceinline__ uint32_t wave_min_pk(uint32_t x)
{
    x = pk_min_s(x, dpp_view<DPP_QUAD_1032>(x));
    x = pk_min_s(x, dpp_view<DPP_QUAD_2301>(x));
    x = pk_min_s(x, dpp_view<DPP_ROW_HALF_MIRROR>(x));
    x = pk_min_s(x, dpp_view<DPP_ROW_MIRROR>(x));
    x = pk_min_s(x, dpp_view<DPP_ROW_BCAST15, 0xa>(x));
    x = pk_min_s(x, dpp_view<DPP_ROW_BCAST31, 0xc>(x));
    return __builtin_amdgcn_readlane(x, 63);
}
// min of the two halves of a wave-uniform packed pair
__device__ __forceinline__ uint32_t halves_min(uint32_t pk) { return min(pk & 0xffffu, pk >> 16); }
// {a.lo, b.lo} and {a.hi, b.hi}
__device__ __forceinline__ uint32_t pack_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- minima of several directions at once, returned SPLAT ({m, m}) in SGPRs ------------------------
// Operands: per-lane packed partial minima of values in [0, 0x7fff] (inside the int16 regime every
// path cost is; outside it hr[0] of the headroom record says so whatever these return).  Once both
// halves of a dword are equal, the unsigned 32-bit order of such dwords is the order of the value, so
// a butterfly step is ONE v_min_u32 with a DPP operand instead of v_mov_b32_dpp + v_pk_min_i16.
// gfx950's half / row swaps (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the
// second; v_permlane16_swap: odd rows of the first <-> even rows of the second) fold four independent
// reductions into one register in three steps: 12 vector instructions + 4 v_readlane for the four
// directions of a sweep pixel (before: 2 perm-packs of 3, two packed butterflies of 12, 2 readlanes and
// the scalar unpack / splat), and the result is the splat the normalisation subtracts.
__device__ __forceinline__ uint32_t fold_halves(uint32_t x) { return pk_min_s(x, __builtin_amdgcn_alignbit(x, x, 16)); }
// minimum over each row of 16 lanes, left in every lane of the row (equal-halves dwords)
__device__ __forceinline__ uint32_t row_min_eq(uint32_t x)
{
    x = min(x, dpp_view<DPP_QUAD_1032>(x));
    x = min(x, dpp_view<DPP_QUAD_2301>(x));
    x = min(x, dpp_view<DPP_ROW_HALF_MIRROR>(x));
    x = min(x, dpp_view<DPP_ROW_MIRROR>(x));
    return x;
}
// lanes 0-31: min over the wave of a, lanes 32-63: of b (per lane pair l, l + 32)
__device__ __forceinline__ uint32_t fold32_pair(uint32_t a, uint32_t b)
{
    const auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return pk_min_s(s[0], s[1]);
}
// rows 0 / 2: rows 0-1 / 2-3 of a folded, rows 1 / 3: rows 0-1 / 2-3 of b folded
__device__ __forceinline__ uint32_t fold16_pair(uint32_t a, uint32_t b)
{
    const auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return pk_min_s(s[0], s[1]);
}
// `rows` (out): the register the four splats are read from -- every lane of row r holds its direction's {m, m}; callers
// that only want the largest of many such minima (the headroom record) keep a per-lane running maximum of it (ONE vector
// instruction per pixel; unsigned 32-bit order = order of the value, as above) instead of four scalar maxima.
__device__ __forceinline__ void wave_min4_splat(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t (&ms)[4], uint32_t &rows)
{
    // rows of 16 lanes after the two folds: directions 0, 2, 1, 3
    const uint32_t x = row_min_eq(fold_halves(fold16_pair(fold32_pair(r0, r1), fold32_pair(r2, r3))));
    rows = x;
    ms[0] = __builtin_amdgcn_readlane(x, 0);
    ms[2] = __builtin_amdgcn_readlane(x, 16);
    ms[1] = __builtin_amdgcn_readlane(x, 32);
    ms[3] = __builtin_amdgcn_readlane(x, 48);
}
__device__ __forceinline__ void wave_min3_splat(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t (&ms)[3])
{
    // rows: direction 0, rows 0-1 of direction 2, direction 1, rows 2-3 of direction 2
    const uint32_t x = row_min_eq(fold_halves(fold16_pair(fold32_pair(r0, r1), r2)));
    ms[0] = __builtin_amdgcn_readlane(x, 0);
    ms[1] = __builtin_amdgcn_readlane(x, 32);
    ms[2] = min(__builtin_amdgcn_readlane(x, 16), __builtin_amdgcn_readlane(x, 48));
}
__device__ __forceinline__ uint32_t wave_min1_splat(uint32_t r)
{
    uint32_t x = row_min_eq(fold_halves(r));
    x = min(x, dpp_view<DPP_ROW_BCAST15, 0xa>(x));
    x = min(x, dpp_view<DPP_ROW_BCAST31, 0xc>(x));
    return __builtin_amdgcn_readlane(x, 63);
}

// min over the 64 lanes of an unsigned 32-bit value; result is wave-uniform (SGPR)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    x = min(x, dpp_view<DPP_QUAD_1032>(x));
    x = min(x, dpp_view<DPP_QUAD_2301>(x));
    x = min(x, dpp_view<DPP_ROW_HALF_MIRROR>(x));
    x = min(x, dpp_view<DPP_ROW_MIRROR>(x));
    x = min(x, dpp_view<DPP_ROW_BCAST15, 0xa>(x));
    x = min(x, dpp_view<DPP_ROW_BCAST31, 0xc>(x));
    return __builtin_amdgcn_readlane(x, 63);
}

// max over the 64 lanes of an unsigned 32-bit value (headroom records: once per wave, at its end)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
    x = max(x, dpp_view<DPP_QUAD_1032>(x));
    x = max(x, dpp_view<DPP_QUAD_2301>(x));
    x = max(x, dpp_view<DPP_ROW_HALF_MIRROR>(x));
    x = max(x, dpp_view<DPP_ROW_MIRROR>(x));
    x = max(x, dpp_view<DPP_ROW_BCAST15, 0xa>(x));
    x = max(x, dpp_view<DPP_ROW_BCAST31, 0xc>(x));
    return __builtin_amdgcn_readlane(x, 63);
}
// Headroom record of one compute (sgm_get_headroom): hr[0] = max of C_true (incl. the running-sum
// intermediate C(y-1) + hsum(y+r) upstream holds in an int16 lane), hr[1] = max over pixels and
// directions of min_d L_r(p, d).  `v` holds per-lane packed uint16 maxima; one atomic per wave.
// One wave's contribution to a headroom maximum (call from ONE lane).  Thousands of waves end a kernel
// with an atomic on the same address; same-address atomics are serialised at the memory side
// (MI355X_MICROARCH.md: "every workgroup into ONE row: 14x slower") -- 3584 of them kept a pre-pass chunk
// of 8 rows alive for 45 us.  The record only grows, so a plain read first is safe in both directions
// (a stale, smaller value merely costs the atomic), and after the first few waves nearly all skip it.
__device__ __forceinline__ void headroom_raise(uint32_t *p, uint32_t m)
{
    if (m > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, m);
}
__device__ __forceinline__ void headroom_commit_pk(uint32_t *hr, int slot, uint32_t v)
{
    const uint32_t m = wave_max_u32(max(v & 0xffffu, v >> 16));
    if (hr && lane_id() == 0) headroom_raise(hr + slot, m);
}

// acc = max(acc, x) on the scalar unit (both wave-uniform).  Written as asm: left to the compiler a chain
// of such maxima turns into v_max3_u32 in the vector stream of the issue-bound kernels that use it.
__device__ __forceinline__ void smax_u32(uint32_t &acc, uint32_t x)
{
    // (s_max_u32 writes SCC: it must be declared, or the compiler keeps a compare result across the statement)
    asm("s_max_u32 %0, %0, %1" : "+s"(acc) : "s"(__builtin_amdgcn_readfirstlane(x)) : "scc");
}

// N independent reductions with their DPP steps interleaved: a DPP read needs two wait states
// after the VALU write of its source, which a second chain fills (a lone chain gets s_nop's).
template <int N> __device__ __forceinline__ void wave_min_pk_n(uint32_t (&x)[N])
{
#define SGM_STEP(CTRL, MASK)                                                        \
    _Pragma("unroll") for (int n = 0; n < N; n++) x[n] = pk_min_s(x[n], dpp_view<CTRL, MASK>(x[n]));
    SGM_STEP(DPP_QUAD_1032, 0xf)
    SGM_STEP(DPP_QUAD_2301, 0xf)
    SGM_STEP(DPP_ROW_HALF_MIRROR, 0xf)
    SGM_STEP(DPP_ROW_MIRROR, 0xf)
    SGM_STEP(DPP_ROW_BCAST15, 0xa)
    SGM_STEP(DPP_ROW_BCAST31, 0xc)
#undef SGM_STEP
#pragma unroll
    for (int n = 0; n < N; n++) x[n] = __builtin_amdgcn_readlane(x[n], 63);
}
template <int N> __device__ __forceinline__ void wave_min_u32_n(uint32_t (&x)[N])
{
#define SGM_STEP(CTRL, MASK)                                                        \
    _Pragma("unroll") for (int n = 0; n < N; n++) x[n] = min(x[n], dpp_view<CTRL, MASK>(x[n]));
    SGM_STEP(DPP_QUAD_1032, 0xf)
    SGM_STEP(DPP_QUAD_2301, 0xf)
    SGM_STEP(DPP_ROW_HALF_MIRROR, 0xf)
    SGM_STEP(DPP_ROW_MIRROR, 0xf)
    SGM_STEP(DPP_ROW_BCAST15, 0xa)
    SGM_STEP(DPP_ROW_BCAST31, 0xc)
#undef SGM_STEP
#pragma unroll
    for (int n = 0; n < N; n++) x[n] = __builtin_amdgcn_readlane(x[n], 63);
}

// ---- lane groups (small D): a wave of 64 lanes holds 64/GW independent pixels of GW lanes each ----
// Reductions that leave the group's result in EVERY lane of the group (a VGPR value that differs
// between groups); GW = 64 is the wave-uniform case above.
template <int GW, int N> __device__ __forceinline__ void group_min_pk_n(uint32_t (&x)[N])
{
    if constexpr (GW == 64) {
        wave_min_pk_n<N>(x);
    } else {
#define SGM_STEP(CTRL, MASK) \
    _Pragma("unroll") for (int n = 0; n < N; n++) x[n] = pk_min_s(x[n], dpp_view<CTRL, MASK>(x[n]));
        SGM_STEP(DPP_QUAD_1032, 0xf)
        SGM_STEP(DPP_QUAD_2301, 0xf)
        SGM_STEP(DPP_ROW_HALF_MIRROR, 0xf)  // 8 lanes done
        if constexpr (GW >= 16) { SGM_STEP(DPP_ROW_MIRROR, 0xf) }
        if constexpr (GW == 32) {
            SGM_STEP(DPP_ROW_BCAST15, 0xa)  // lanes 31 / 63 now hold the minimum of rows 0-1 / 2-3
            const bool lower = lane_id() < 32;
#pragma unroll
            for (int n = 0; n < N; n++) {
                const uint32_t a = __builtin_amdgcn_readlane(x[n], 31), b = __builtin_amdgcn_readlane(x[n], 63);
                x[n] = lower ? a : b;
            }
        }
#undef SGM_STEP
    }
}
template <int GW, int N> __device__ __forceinline__ void group_min_u32_n(uint32_t (&x)[N])
{
    if constexpr (GW == 64) {
        wave_min_u32_n<N>(x);
    } else {
#define SGM_STEP(CTRL, MASK) \
    _Pragma("unroll") for (int n = 0; n < N; n++) x[n] = min(x[n], dpp_view<CTRL, MASK>(x[n]));
        SGM_STEP(DPP_QUAD_1032, 0xf)
        SGM_STEP(DPP_QUAD_2301, 0xf)
        SGM_STEP(DPP_ROW_HALF_MIRROR, 0xf)
        if constexpr (GW >= 16) { SGM_STEP(DPP_ROW_MIRROR, 0xf) }
        if constexpr (GW == 32) {
            SGM_STEP(DPP_ROW_BCAST15, 0xa)
            const bool lower = lane_id() < 32;
#pragma unroll
            for (int n = 0; n < N; n++) {
                const uint32_t a = __builtin_amdgcn_readlane(x[n], 31), b = __builtin_amdgcn_readlane(x[n], 63);
                x[n] = lower ? a : b;
            }
        }
#undef SGM_STEP
    }
}
template <int GW> __device__ __forceinline__ uint32_t group_min_pk(uint32_t x)
{
    uint32_t a[1] = {x};
    group_min_pk_n<GW, 1>(a);
    return a[0];
}

// One direction's minimum over a lane group, returned as the splat {m, m} in every lane of the group
// (GW = 64: wave-uniform).  Halves folded first, then one v_min_u32 with a DPP operand per butterfly step:
// 2 + log2(GW) dependent instructions instead of 2 * log2(GW) + the scalar halves-minimum and splat --
// the in-row kernels are bound by exactly this chain.  Operands in [0, 0x7fff] (see wave_min4_splat).
template <int GW> __device__ __forceinline__ uint32_t group_min_splat(uint32_t r)
{
    if constexpr (GW == 64) {
        return wave_min1_splat(r);
    } else {
        uint32_t x = fold_halves(r);
        x = min(x, dpp_view<DPP_QUAD_1032>(x));
        x = min(x, dpp_view<DPP_QUAD_2301>(x));
        x = min(x, dpp_view<DPP_ROW_HALF_MIRROR>(x));  // 8 lanes done
        if constexpr (GW >= 16) x = min(x, dpp_view<DPP_ROW_MIRROR>(x));
        // two rows per group: every lane holds its row's minimum; v_permlane16_swap pairs row 0 with 1 and 2 with 3
        // (copy + swap + minimum instead of row_bcast + minimum + two v_readlane + two moves + a select)
        if constexpr (GW == 32) x = fold16_pair(x, x);
        return x;
    }
}

// Volume and boundary-state stores are streaming ("nt"): nothing written by a kernel is read again by the
// same kernel, and without the hint the written lines push the lines that ARE re-read (the byte costs
// under k_box_u8's window, C under the three roles of the pre-pass) out of the 4 MiB L2s.  Measured on
// the 4K MODE_HH frame: k_box_u8 1.152 -> 1.126 ms, pre-pass 1.668 -> 1.633 ms, frame -0.5 ... -1.5 %.
// (-DSGM_NT_STORES=0 builds the plain-store variant for A/B runs through SGM_HIP_LIB.)
#ifndef SGM_NT_STORES
#define SGM_NT_STORES 1
#endif

// NP packed registers per lane, moved as one vector access
template <int NP> struct PackVec;
template <> struct PackVec<1> { typedef uint32_t type; };
template <> struct PackVec<2> { typedef uint2 type; };
template <> struct PackVec<4> { typedef uint4 type; };

template <int NP> struct Pack {
    uint32_t r[NP];
    __device__ __forceinline__ void load(const int16_t *p)
    {
        typename PackVec<NP>::type v = *reinterpret_cast<const typename PackVec<NP>::type *>(p);
        __builtin_memcpy(r, &v, sizeof(v));
    }
    __device__ __forceinline__ void store(int16_t *p) const
    {
        typename PackVec<NP>::type v;
        __builtin_memcpy(&v, r, sizeof(v));
#if SGM_NT_STORES
        typedef unsigned nt_vec __attribute__((ext_vector_type(NP)));
        nt_vec w;
        __builtin_memcpy(&w, r, sizeof(w));
        if constexpr (NP == 1) __builtin_nontemporal_store(r[0], reinterpret_cast<uint32_t *>(p));
        else __builtin_nontemporal_store(w, reinterpret_cast<nt_vec *>(p));
#else
        *reinterpret_cast<typename PackVec<NP>::type *>(p) = v;
#endif
    }
    __device__ __forceinline__ void fill(uint32_t x)
    {
#pragma unroll
        for (int i = 0; i < NP; i++) r[i] = x;
    }
};


// Buffer (SRSRC) loads: one wave-uniform descriptor + one per-lane 32-bit offset register that
// never changes + a scalar byte offset per access.  Used where many loads are issued back to back
// (the sweep's loader wave): with flat/global addressing hipcc gave every access its own VGPR
// offset, reused those registers as load destinations and had to drain vmcnt(0) inside the burst.
#if SGM_NT_STORES
#define SGM_ST_AUX 2  // nt
#else
#define SGM_ST_AUX 0
#endif
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
template <int NP, int AUX = 0>
__device__ __forceinline__ void buf_load(Pack<NP> &p, __amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes)
{
    if constexpr (NP == 1) {
        p.r[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_bytes, soff_bytes, AUX);
    } else if constexpr (NP == 2) {
        const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_bytes, soff_bytes, AUX);
        p.r[0] = v.x;
        p.r[1] = v.y;
    } else {
        const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, AUX);
        p.r[0] = v.x;
        p.r[1] = v.y;
        p.r[2] = v.z;
        p.r[3] = v.w;
    }
}
// AUX: cache policy bits of the store (2 = nt: streaming, the default; 16 = sc1: write-through, agent scope --
// the chained sweep's hand-off records)
template <int NP, int AUX = SGM_ST_AUX>
__device__ __forceinline__ void buf_store(const Pack<NP> &p, __amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes)
{
    if constexpr (NP == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(p.r[0], rsrc, voff_bytes, soff_bytes, AUX);
    } else if constexpr (NP == 2) {
        v2u32 v;
        v.x = p.r[0];
        v.y = p.r[1];
        __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voff_bytes, soff_bytes, AUX);
#ifdef SGM_EXPERIMENT_STORE_X4  // ISA study only; never defined in a shipped build
    } else if constexpr (NP == 4) {
        v4u32 v;
        v.x = p.r[0];
        v.y = p.r[1];
        v.z = p.r[2];
        v.w = p.r[3];
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff_bytes, soff_bytes, AUX);
#endif
    } else {
        // Two 64-bit stores, never buffer_store_dwordx4 (root cause found in round 2).
        // A MUBUF store of more than 64 bits reads its upper data registers a cycle or two after
        // issue: a VALU write to them needs 1-2 wait states behind the store.  hipcc (ROCm 7.2,
        // GCNHazardRecognizer::createsVALUHazard) pads that hazard only when soffset is an
        // immediate; with an SGPR soffset -- every store here -- it pads nothing, and in k_sweep it
        // scheduled "buffer_store_dwordx4 v[12:15], .., s35 offen" directly in front of
        // "v_perm_b32 v12, .." (the first instruction of the next min-reduction).  Under memory
        // back-pressure the store then picked up the new v12 in the last quad of every row of 16
        // lanes: exactly the wrong S dwords seen in round 1 (the stored value is the per-lane
        // partial minimum the reduction starts from).  64-bit stores read all their data at issue
        // and have no such hazard.  The empty asm keeps the two halves from being merged again;
        // tests/test_abi.py checks the generated ISA for buffer_store_dwordx3/x4.
        v2u32 lo, hi;
        lo.x = p.r[0];
        lo.y = p.r[1];
        hi.x = p.r[2];
        hi.y = p.r[3];
        __builtin_amdgcn_raw_buffer_store_b64(lo, rsrc, voff_bytes, soff_bytes, AUX);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_raw_buffer_store_b64(hi, rsrc, voff_bytes + 8, soff_bytes, AUX);
    }
}

// Buffer descriptor from values the compiler must be able to PROVE wave-uniform (guide T20): a
// descriptor whose words it takes for divergent -- anything derived from threadIdx, even lane / 64,
// or from an integer division it expands in the vector unit, or merely defined behind a divergent
// early return -- lives in VGPRs, and every buffer load / store through it is wrapped in a
// "waterfall" loop (4 v_readfirstlane + 2 v_cmp + s_and_saveexec + branch, one memory operation at a
// time).  Round 2 found k_box_u8, k_pix, k_hsum and the one-row-per-wave k_rows_g built that way.
// v_readfirstlane on the base pointer and the byte count costs three instructions once per wave.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base, int64_t byte_off, int bytes)
{
    const uint64_t u = (uint64_t)base + (uint64_t)byte_off;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
}

#define SGM_MAX_COST 32767
#define SGM_SENT 0x7fff7fffu  // packed pair of MAX_COST

}  // namespace sgm

/*
 * dev_prims.h -- the handful of wave-level primitives the kernels are written against.
 * Device build (hipcc, gfx950): thin wrappers over CDNA4 builtins; wave = 64 lanes.
 * FPL_EMU build (g++, tests only): the same names on top of tests/emu/hip_emu.h.
 */
#ifndef FPL_DEV_PRIMS_H
#define FPL_DEV_PRIMS_H

#include <stdint.h>

#ifdef FPL_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

#ifdef FPL_EMU
#define FPL_NOINLINE
#else
#define FPL_NOINLINE __attribute__((noinline))
#endif

namespace fpl {

constexpr int WAVE = 64;

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

struct u32x4 {
    u32 x, y, z, w;
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
/* index of this wave in its block, in a scalar register (every lane of a wave has the same value) */
__device__ __forceinline__ int wave_in_block() {
#ifdef FPL_EMU
    return (int)(threadIdx.x >> 6);
#else
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#endif
}

__device__ __forceinline__ u64 wave_ballot(bool p) { return __ballot(p ? 1 : 0); }

/* hide where a value came from (no instruction): keeps the optimiser from rewriting an expression around it */
__device__ __forceinline__ void opaque_u32(u32& v) {
#ifndef FPL_EMU
    asm volatile("" : "+v"(v));
#else
    (void)v;
#endif
}

/* order this wave's LDS traffic across lanes (zero -> atomics -> reads of a per-wave table) */
__device__ __forceinline__ void wave_sync() {
#ifdef FPL_EMU
    emu_wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

__device__ __forceinline__ u32 shfl_u32(u32 v, int src) { return (u32)__shfl((int)v, src, 64); }
__device__ __forceinline__ int shfl_i32(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ u32 shfl_up_u32(u32 v, unsigned d) { return (u32)__shfl_up((int)v, d, 64); }
__device__ __forceinline__ u32 shfl_down_u32(u32 v, unsigned d) { return (u32)__shfl_down((int)v, d, 64); }
__device__ __forceinline__ u32 shfl_xor_u32(u32 v, int m) { return (u32)__shfl_xor((int)v, m, 64); }
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    u32 lo = shfl_u32((u32)v, src), hi = shfl_u32((u32)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    u32 lo = shfl_xor_u32((u32)v, m), hi = shfl_xor_u32((u32)(v >> 32), m);
    return ((u64)hi << 32) | lo;
}

/* value of lane `src` (wave-uniform index) in every lane: v_readlane_b32, no LDS traffic */
__device__ __forceinline__ u32 readlane_u32(u32 v, int src) {
#ifdef FPL_EMU
    return shfl_u32(v, src);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, src);
#endif
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int src) {
    return ((u64)readlane_u32((u32)(v >> 32), src) << 32) | readlane_u32((u32)v, src);
}
/* a per-lane value every lane may read at wave-uniform indices: v_readlane on the device; the
 * emulator snapshots all 64 values once instead of paying a rendezvous per read */
struct WaveVals64 {
#ifdef FPL_EMU
    u64 vals[64];
    u64 get(int t) const { return vals[t]; }
#else
    u64 v;
    __device__ __forceinline__ u64 get(int t) const { return readlane_u64(v, t); }
#endif
};
__device__ __forceinline__ WaveVals64 wave_publish(u64 v) {
    WaveVals64 w;
#ifdef FPL_EMU
    emu_gather_u64(v, w.vals);
#else
    w.v = v;
#endif
    return w;
}

/* wave-wide reductions / scans (all 64 lanes must call, in wave-uniform control flow).
 * Device: DPP row shifts and row broadcasts (one VALU op per step, no LDS crossbar trip like
 * ds_bpermute): row_shr 1,2,4,8 scan each row of 16 lanes, row_bcast:15 / row_bcast:31 carry the row
 * totals forward; lane 63 ends up with the reduction of the whole wave.  Lanes without a source keep
 * `old`, the identity of the operation. */
#ifndef FPL_EMU
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_u32(u32 old, u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
#define FPL_DPP_SCAN(OP, ID, v)                          \
    v = OP(v, (dpp_u32<0x111, 0xf>(ID, v)));             \
    v = OP(v, (dpp_u32<0x112, 0xf>(ID, v)));             \
    v = OP(v, (dpp_u32<0x114, 0xf>(ID, v)));             \
    v = OP(v, (dpp_u32<0x118, 0xf>(ID, v)));             \
    v = OP(v, (dpp_u32<0x142, 0xa>(ID, v)));             \
    v = OP(v, (dpp_u32<0x143, 0xc>(ID, v)));
__device__ __forceinline__ u32 op_add_u32(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 op_max_u32(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 op_min_u32(u32 a, u32 b) { return a < b ? a : b; }
#endif
/* the value of the next lane (lane 63 gets 0): DPP wave_shl:1, no LDS */
__device__ __forceinline__ u32 wave_next_u32(u32 v) {
#ifdef FPL_EMU
    const u32 o = shfl_down_u32(v, 1);
    return lane_id() == 63 ? 0u : o;
#else
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
#endif
}
/* the value of the previous lane (lane 0 gets `first`): DPP wave_shr:1, no LDS */
__device__ __forceinline__ u32 wave_prev_u32(u32 v, u32 first) {
#ifdef FPL_EMU
    const u32 o = shfl_up_u32(v, 1);
    return lane_id() == 0 ? first : o;
#else
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
#endif
}
/* inclusive prefix sum across lanes */
__device__ __forceinline__ u32 wave_scan_incl_u32(u32 v) {
#ifdef FPL_EMU
    int l = lane_id();
    for (int d = 1; d < 64; d <<= 1) {
        u32 o = shfl_up_u32(v, d);
        if (l >= d) v += o;
    }
    return v;
#else
    FPL_DPP_SCAN(op_add_u32, 0u, v)
    return v;
#endif
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#ifdef FPL_EMU
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u32(v, m);
    return v;
#else
    FPL_DPP_SCAN(op_add_u32, 0u, v)
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
#endif
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#ifdef FPL_EMU
    for (int m = 32; m >= 1; m >>= 1) {
        u32 o = shfl_xor_u32(v, m);
        v = o > v ? o : v;
    }
    return v;
#else
    FPL_DPP_SCAN(op_max_u32, 0u, v)
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
#endif
}
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#ifdef FPL_EMU
    for (int m = 32; m >= 1; m >>= 1) {
        u32 o = shfl_xor_u32(v, m);
        v = o < v ? o : v;
    }
    return v;
#else
    FPL_DPP_SCAN(op_min_u32, ~0u, v)
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
#endif
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#ifdef FPL_EMU
    for (int m = 32; m >= 1; m >>= 1) {
        u64 o = shfl_xor_u64(v, m);
        v = o < v ? o : v;
    }
    return v;
#else
#define FPL_DPP_MIN64(CTRL, MASK)                                                                       \
    {                                                                                                   \
        const u64 o = ((u64)dpp_u32<CTRL, MASK>(~0u, (u32)(v >> 32)) << 32) | dpp_u32<CTRL, MASK>(~0u, (u32)v); \
        v = o < v ? o : v;                                                                              \
    }
    FPL_DPP_MIN64(0x111, 0xf)
    FPL_DPP_MIN64(0x112, 0xf)
    FPL_DPP_MIN64(0x114, 0xf)
    FPL_DPP_MIN64(0x118, 0xf)
    FPL_DPP_MIN64(0x142, 0xa)
    FPL_DPP_MIN64(0x143, 0xc)
#undef FPL_DPP_MIN64
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, 63);
#endif
}

/* bytes [n, n+4) of the 8-byte little-endian value hi:lo, n in 0..3 */
__device__ __forceinline__ u32 alignbyte(u32 hi, u32 lo, u32 n) {
#ifdef FPL_EMU
    return (u32)(((((u64)hi) << 32) | lo) >> (8 * (n & 3)));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, n);
#endif
}

/* bits [n, n+32) of the 64-bit value hi:lo, n in 0..31 */
__device__ __forceinline__ u32 alignbit(u32 hi, u32 lo, u32 n) {
#ifdef FPL_EMU
    return (u32)(((((u64)hi) << 32) | lo) >> (n & 31));
#else
    return __builtin_amdgcn_alignbit(hi, lo, n);
#endif
}

/* the bits of x in reverse order (v_bfrev_b32) */
__device__ __forceinline__ u32 brev32(u32 x) {
#ifdef FPL_EMU
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#else
    return __builtin_bitreverse32(x);
#endif
}

/* any three-input bitwise function in one VALU op (v_bitop3_b32); TT = f(0xF0, 0xCC, 0xAA) */
template <int TT>
__device__ __forceinline__ u32 bitop3(u32 a, u32 b, u32 c) {
#ifdef FPL_EMU
    u32 r = 0;
    for (int i = 0; i < 32; i++) {
        const int idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1);
        r |= (u32)((TT >> idx) & 1) << i;
    }
    return r;
#else
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#endif
}
/* carry-save adder on bit-planes: (h, l) = a + b + c */
__device__ __forceinline__ void csa(u32& h, u32& l, u32 a, u32 b, u32 c) {
    const u32 hh = bitop3<0xE8>(a, b, c); /* majority */
    l = bitop3<0x96>(a, b, c);            /* parity   */
    h = hh;
}

/* sum over the 4 bytes of a.byte * b.byte, plus c (v_dot4_u32_u8) */
__device__ __forceinline__ u32 udot4(u32 a, u32 b, u32 c) {
#ifdef FPL_EMU
    u32 r = c;
    for (int i = 0; i < 4; i++) r += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return r;
#else
    return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
/* Wait states behind dot products whose results the NEXT instructions use.  Measured on the MI355X (round 4): a vector
   instruction that reads a v_dot4 result one instruction behind the dot gets the register's old value -- the rule LLVM's
   hazard recogniser has for gfx90a (a different instruction reads a dot's destination: 3 wait states, overwrites it: 4) holds
   on gfx950 too, and this hipcc does not insert the nops.  A dot that feeds the next dot's accumulator is forwarded and needs
   none.  tools/dot_hazard_isa.py checks every v_dot4 of the built library (tests/test_isa_dequeue.py). */
__device__ __forceinline__ void dot_settle(u32& a, u32& b) {
#ifndef FPL_EMU
    asm volatile("s_nop 3" : "+v"(a), "+v"(b));
#else
    (void)a;
    (void)b;
#endif
}
/* sum of the 4 bytes of a, plus c (v_sad_u8 against 0) */
__device__ __forceinline__ u32 sum_bytes(u32 a, u32 c) {
#ifdef FPL_EMU
    return c + (a & 0xFF) + ((a >> 8) & 0xFF) + ((a >> 16) & 0xFF) + (a >> 24);
#else
    return __builtin_amdgcn_sad_u8(a, 0u, c);
#endif
}
/* result byte i = byte sel.byte[i] (0..3) of tbl (v_perm_b32 with a zero high half) */
__device__ __forceinline__ u32 perm_lo(u32 tbl, u32 sel) {
#ifdef FPL_EMU
    u32 r = 0;
    for (int i = 0; i < 4; i++) {
        const u32 k = (sel >> (8 * i)) & 0xFF;
        r |= ((k < 4 ? (tbl >> (8 * k)) : 0) & 0xFF) << (8 * i);
    }
    return r;
#else
    return __builtin_amdgcn_perm(0u, tbl, sel);
#endif
}
/* v_perm_b32: result byte i = byte sel.byte[i] of the 8 bytes hi:lo (0..3 lo, 4..7 hi), 0x0c = constant 0 */
__device__ __forceinline__ u32 perm_b32(u32 hi, u32 lo, u32 sel) {
#ifdef FPL_EMU
    const u64 v = ((u64)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; i++) {
        const u32 k = (sel >> (8 * i)) & 0xFF;
        r |= (u32)((k < 8 ? (v >> (8 * k)) : 0) & 0xFF) << (8 * i);
    }
    return r;
#else
    return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}
/* a value the program knows to be wave-uniform, moved to a scalar register so that everything derived from
   it runs on the scalar unit */
__device__ __forceinline__ u32 uniform_u32(u32 v) {
#ifdef FPL_EMU
    return v;
#else
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}
__device__ __forceinline__ int uniform_i32(int v) { return (int)uniform_u32((u32)v); }
__device__ __forceinline__ u64 uniform_u64(u64 v) {
    return ((u64)uniform_u32((u32)(v >> 32)) << 32) | uniform_u32((u32)v);
}
__device__ __forceinline__ int readlane_i32(int v, int src) { return (int)readlane_u32((u32)v, src); }
/* (a << n) | b in one VALU op (v_lshl_or_b32; spelled out so that the compiler does not turn a chain of
   them into a quarter-rate 32-bit multiply) */
template <int N>
__device__ __forceinline__ u32 lshl_or(u32 a, u32 b) {
#ifdef FPL_EMU
    return (a << N) | b;
#else
    u32 r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(N), "v"(b));
    return r;
#endif
}
/* a * b + c for a, b < 2^24 in one full-rate VALU op (v_mad_u32_u24) */
__device__ __forceinline__ u32 mad_u24(u32 a, u32 b, u32 c) {
#ifdef FPL_EMU
    return a * b + c;
#else
    u32 r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
#endif
}
__device__ __forceinline__ u32 popc32(u32 x) { return (u32)__popc(x); }
/* popcount(x) + acc in one VALU op (v_bcnt_u32_b32 takes an addend; left to itself the compiler counts against zero
   and sums the counts with v_add3_u32: three instructions for two counts instead of two) */
__device__ __forceinline__ u32 popc_acc(u32 x, u32 acc) {
#ifdef FPL_EMU
    return (u32)__popc(x) + acc;
#else
    u32 r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
#endif
}
/* (a & b) | c in one VALU op (v_and_or_b32; b wave-uniform) */
__device__ __forceinline__ u32 and_or(u32 a, u32 b, u32 c) {
#ifdef FPL_EMU
    return (a & b) | c;
#else
    u32 r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
#endif
}

/* 0x01 in every byte of x that is non-zero */
__device__ __forceinline__ u32 nonzero_bytes01(u32 x) {
    u32 y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ((y | x) >> 7) & 0x01010101u;
}

/* 16-byte load from an arbitrarily aligned address (gfx950 global loads take any
 * alignment; the compiler emits one global_load_dwordx4) */
__device__ __forceinline__ u32x4 load16(const u8* p) {
    u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
/* guarded variant: bytes at or beyond `end` read as 0 (the slow tail path is kept out of line) */
__device__ FPL_NOINLINE u32x4 load16_tail(const u8* p, const u8* end) {
    u32 w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++)
        if (p + i < end) w[i >> 2] |= (u32)p[i] << (8 * (i & 3));
    u32x4 v = {w[0], w[1], w[2], w[3]};
    return v;
}
__device__ __forceinline__ u32x4 load16_guard(const u8* p, const u8* end) {
    if (p + 16 <= end) return load16(p);
    return load16_tail(p, end);
}
__device__ __forceinline__ u32 load4_guard(const u8* p, const u8* end) {
    u32 w = 0;
    if (p + 4 <= end) {
        __builtin_memcpy(&w, p, 4);
        return w;
    }
    for (int i = 0; i < 4; i++)
        if (p + i < end) w |= (u32)p[i] << (8 * i);
    return w;
}

}  // namespace fpl
#endif

/*
 * prims_test.hip -- TEST INFRASTRUCTURE ONLY (tests/gpu_prims/libprims_test.so, built by tests/gpu_prims/build.py).
 *
 * Unit kernels for the hand-written primitives the hot kernels stand on and that the CPU emulator cannot see, because it
 * compiles their C fallbacks instead (FPL_EMU): sliced_max's add-with-carry form (inline asm: v_addc_co_u32 fed by a ballot
 * pair, a manual s_nop for the VALU -> SGPR hazard), wave_prev_u32 (DPP wave_shr:1), the DPP reductions and scans.  Every
 * case is computed twice on the GPU -- by the product function and by a plain per-lane loop over shared memory -- and both
 * results go back to the host; tests/test_gpu_prims.py compares them with each other and with numpy.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../fastplong_amd/csrc/kernels.h"

using namespace fpl;

/* in:  [case][lane][8] = 7 count planes + the candidate mask; act[case] = the lanes that take part (bit per lane)
   out: [case][lane][4] = val, first by sliced_max<NB>; val, first by the plain loop */
template <int NB>
__global__ void k_sliced_max(const u32* __restrict__ in, const unsigned long long* __restrict__ act, u32* __restrict__ out) {
    const int lane = lane_id();
    const u32* my = in + ((size_t)blockIdx.x * 64 + lane) * 8;
    u32 B[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) B[b] = my[b];
    const u32 cand = my[7];
    u32* o = out + ((size_t)blockIdx.x * 64 + lane) * 4;
    o[0] = o[1] = o[2] = o[3] = 0xDEADBEEFu;
    if (((act[blockIdx.x] >> lane) & 1ull) && cand) { /* partial exec mask: the lanes left out must not disturb the others */
        int val, first;
        sliced_max<NB>(B, cand, val, first);
        o[0] = (u32)val;
        o[1] = (u32)first;
        int rv = -1, rf = -1;
        for (int p = 0; p < 32; p++) {
            if (!((cand >> p) & 1u)) continue;
            int c = 0;
            for (int b = 0; b < NB; b++) c |= (int)((B[b] >> p) & 1u) << b;
            if (c > rv) {
                rv = c;
                rf = p;
            }
        }
        o[2] = (u32)rv;
        o[3] = (u32)rf;
    }
}

/* out: [case][lane][8] = wave_prev_u32, wave_sum_u32, wave_scan_incl_u32, wave_max_u32, wave_min_u32, wave_min_u64 (lo, hi), readlane(v, 37) */
__global__ void k_wave_prims(const u32* __restrict__ in, u32* __restrict__ out) {
    const int lane = lane_id();
    const u32 v = in[(size_t)blockIdx.x * 64 + lane];
    u32* o = out + ((size_t)blockIdx.x * 64 + lane) * 8;
    o[0] = wave_prev_u32(v, 0xABCD0000u + blockIdx.x);
    o[1] = wave_sum_u32(v);
    o[2] = wave_scan_incl_u32(v);
    o[3] = wave_max_u32(v);
    o[4] = wave_min_u32(v);
    const u64 k = wave_min_u64(((u64)v << 32) | (u32)(63 - lane));
    o[5] = (u32)k;
    o[6] = (u32)(k >> 32);
    o[7] = readlane_u32(v, 37);
}

/* in: [case][lane][2] = a lane's eight bytes, halo[case] = the four bytes in front of lane 0's
   out: [case][lane][6] = kmer_stream (24 bits), the stream from kmer_pack + v_perm (the form k_stats uses), kmer_pack_dot of
   both dwords, kmer_pack >> 24 of both dwords */
__global__ void k_kmer_stream(const u32* __restrict__ in, const u32* __restrict__ halo, u32* __restrict__ out) {
    const int lane = lane_id();
    const u32 d0 = in[((size_t)blockIdx.x * 64 + lane) * 2], d1 = in[((size_t)blockIdx.x * 64 + lane) * 2 + 1];
    const u32 h = halo[blockIdx.x];
    u32* o = out + ((size_t)blockIdx.x * 64 + lane) * 6;
    const u32 v0 = kmer_codes(d0), v1 = kmer_codes(d1);
    o[0] = kmer_stream(v0, v1, kmer_pack_dot(kmer_codes(h))) & 0xFFFFFFu;
    const u32 up = wave_prev_u32(d1, 0u);
    const u32 vh = kmer_codes(lane > 0 ? up : h);
    o[1] = perm_b32(kmer_pack(vh), perm_b32(kmer_pack(v0), kmer_pack(v1), 0x0c0c0703u), 0x0c070100u);
    o[2] = kmer_pack_dot(v0);
    o[3] = kmer_pack_dot(v1);
    o[4] = kmer_pack(v0) >> 24;
    o[5] = kmer_pack(v1) >> 24;
}

extern "C" {
int prims_kmer_stream(const uint32_t* in, const uint32_t* halo, uint32_t* out, int n_cases) {
    u32 *din = nullptr, *dh = nullptr, *dout = nullptr;
    const size_t ni = (size_t)n_cases * 64 * 2 * 4, no = (size_t)n_cases * 64 * 6 * 4;
    if (hipMalloc((void**)&din, ni) != hipSuccess || hipMalloc((void**)&dout, no) != hipSuccess ||
        hipMalloc((void**)&dh, (size_t)n_cases * 4) != hipSuccess)
        return 1;
    (void)hipMemcpy(din, in, ni, hipMemcpyHostToDevice);
    (void)hipMemcpy(dh, halo, (size_t)n_cases * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_kmer_stream, dim3(n_cases), dim3(64), 0, 0, din, dh, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, no, hipMemcpyDeviceToHost);
    (void)hipFree(din);
    (void)hipFree(dh);
    (void)hipFree(dout);
    return rc;
}
int prims_sliced_max(const uint32_t* in, const unsigned long long* act, uint32_t* out, int n_cases, int nb) {
    u32 *din = nullptr, *dout = nullptr;
    unsigned long long* dact = nullptr;
    const size_t ni = (size_t)n_cases * 64 * 8 * 4, no = (size_t)n_cases * 64 * 4 * 4;
    if (hipMalloc((void**)&din, ni) != hipSuccess || hipMalloc((void**)&dout, no) != hipSuccess ||
        hipMalloc((void**)&dact, (size_t)n_cases * 8) != hipSuccess)
        return 1;
    (void)hipMemcpy(din, in, ni, hipMemcpyHostToDevice);
    (void)hipMemcpy(dact, act, (size_t)n_cases * 8, hipMemcpyHostToDevice);
    if (nb == 6) hipLaunchKernelGGL(k_sliced_max<6>, dim3(n_cases), dim3(64), 0, 0, din, dact, dout);
    else hipLaunchKernelGGL(k_sliced_max<7>, dim3(n_cases), dim3(64), 0, 0, din, dact, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, no, hipMemcpyDeviceToHost);
    (void)hipFree(din);
    (void)hipFree(dout);
    (void)hipFree(dact);
    return rc;
}
int prims_wave(const uint32_t* in, uint32_t* out, int n_cases) {
    u32 *din = nullptr, *dout = nullptr;
    const size_t ni = (size_t)n_cases * 64 * 4, no = (size_t)n_cases * 64 * 8 * 4;
    if (hipMalloc((void**)&din, ni) != hipSuccess || hipMalloc((void**)&dout, no) != hipSuccess) return 1;
    (void)hipMemcpy(din, in, ni, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_wave_prims, dim3(n_cases), dim3(64), 0, 0, din, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, no, hipMemcpyDeviceToHost);
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}
}

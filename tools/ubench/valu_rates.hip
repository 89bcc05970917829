// Profiling aid only (not part of the product): issue rate of the integer / bit VALU, SALU and LDS
// instructions the kernels are made of, on gfx950.  Each kernel runs a long unrolled stream of ONE
// instruction kind over 8 independent register chains; rate = wave-instructions / (ns * CU).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 512;

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

#define KERNEL(NAME, ASM8)                                                                   \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned c0) {                 \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned c = c0 | 1u, d = c0 + 3u;                                                    \
        for (int i = 0; i < ITER; i++) {                                                     \
            asm volatile(ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8 ASM8                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(c), "v"(d), "s"(c0) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                           \
        }                                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;  \
    }

// one instruction per chain; %0..%7 chains, %8 = c (vgpr), %9 = d (vgpr), %10 = sgpr
#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OP3I(op, imm) op " %0, %0, " imm ", %9\n" op " %1, %1, " imm ", %9\n" op " %2, %2, " imm ", %9\n" op " %3, %3, " imm ", %9\n" op " %4, %4, " imm ", %9\n" op " %5, %5, " imm ", %9\n" op " %6, %6, " imm ", %9\n" op " %7, %7, " imm ", %9\n"
#define OPF(a, b, c) a "%0" b "%0" c "\n" a "%1" b "%1" c "\n" a "%2" b "%2" c "\n" a "%3" b "%3" c "\n" a "%4" b "%4" c "\n" a "%5" b "%5" c "\n" a "%6" b "%6" c "\n" a "%7" b "%7" c "\n"
#define OPX(pre, post) pre " %0, %0" post "\n" pre " %1, %1" post "\n" pre " %2, %2" post "\n" pre " %3, %3" post "\n" pre " %4, %4" post "\n" pre " %5, %5" post "\n" pre " %6, %6" post "\n" pre " %7, %7" post "\n"

KERNEL(k_add, OP2("v_add_u32"))
KERNEL(k_and, OP2("v_and_b32"))
KERNEL(k_xor, OP2("v_xor_b32"))
KERNEL(k_lshl, OPF("v_lshlrev_b32 ", ", 3, ", ""))
KERNEL(k_bitop3, OPX("v_bitop3_b32", ", %8, %9 bitop3:0x96"))
KERNEL(k_lshl_or, OP3I("v_lshl_or_b32", "5"))
KERNEL(k_lshl_add, OP3I("v_lshl_add_u32", "2"))
KERNEL(k_and_or, OP3("v_and_or_b32"))
KERNEL(k_or3, OP3("v_or3_b32"))
KERNEL(k_add3, OP3("v_add3_u32"))
KERNEL(k_xad, OP3("v_xad_u32"))
KERNEL(k_bfe, OPX("v_bfe_u32", ", 8, 8"))
KERNEL(k_perm, OP3("v_perm_b32"))
KERNEL(k_alignbit, OP3("v_alignbit_b32"))
KERNEL(k_alignbyte, OP3("v_alignbyte_b32"))
KERNEL(k_mad_u24, OP3("v_mad_u32_u24"))
KERNEL(k_mul_lo, OP2("v_mul_lo_u32"))
KERNEL(k_mul_u24, OP2("v_mul_u32_u24"))
KERNEL(k_dot4, OP3("v_dot4_u32_u8"))
KERNEL(k_sad_u8, OP3("v_sad_u8"))
KERNEL(k_bcnt, OP2("v_bcnt_u32_b32"))
KERNEL(k_min, OP2("v_min_u32"))
KERNEL(k_min3, OP3("v_min3_u32"))
KERNEL(k_cndmask, OPX("v_cndmask_b32", ", %8, vcc"))
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n")
KERNEL(k_addc, "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n")
KERNEL(k_pk_add_u16, OP2("v_pk_add_u16"))
KERNEL(k_pk_lshl_b16, OPF("v_pk_lshlrev_b16 ", ", 1, ", " op_sel_hi:[0,1]"))
KERNEL(k_pk_sub_u16, OP2("v_pk_sub_u16"))
KERNEL(k_sdwa_lshl, OPF("v_lshlrev_b32_sdwa ", ", %8, ", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"))
KERNEL(k_sdwa_add, OPF("v_add_u32_sdwa ", ", ", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD"))
KERNEL(k_dpp_shr, OPF("v_add_u32_dpp ", ", ", ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mov_dpp, OPX("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3\n")
KERNEL(k_add_sgpr, OPF("v_add_u32 ", ", %10, ", ""))
KERNEL(k_salu, "s_add_u32 s20, s20, %10\n s_and_b32 s21, s21, %10\n s_add_u32 s22, s22, %10\n s_xor_b32 s23, s23, %10\n s_add_u32 s24, s24, %10\n s_lshl_b32 s25, s25, 1\n s_add_u32 s26, s26, %10\n s_or_b32 s27, s27, %10\n")
KERNEL(k_mix_valu_salu, "v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, %10\n v_add_u32 %1, %1, %8\n s_and_b32 s21, s21, %10\n v_add_u32 %2, %2, %8\n s_add_u32 s22, s22, %10\n v_add_u32 %3, %3, %8\n s_xor_b32 s23, s23, %10\n")
/* mixed kinds, to see whether different VALU classes co-issue */
KERNEL(k_mix_add_perm, "v_add_u32 %0, %0, %8\n v_perm_b32 %1, %1, %8, %9\n v_add_u32 %2, %2, %8\n v_perm_b32 %3, %3, %8, %9\n v_add_u32 %4, %4, %8\n v_perm_b32 %5, %5, %8, %9\n v_add_u32 %6, %6, %8\n v_perm_b32 %7, %7, %8, %9\n")
KERNEL(k_dep_add, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n")

/* LDS kernels: addresses in a0..a7 stay fixed; data regs separate */
#define LDSKERNEL(NAME, BODY)                                                                \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned c0) {                 \
        __shared__ unsigned long long lds[2048];                                              \
        for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i;                             \
        __syncthreads();                                                                     \
        unsigned base = (unsigned)(size_t)lds;                                                \
        unsigned ad_lin = base + 8u * threadIdx.x;        /* conflict-free, 8-byte stride */  \
        unsigned ad_l4 = base + 4u * threadIdx.x;         /* conflict-free, 4-byte stride */  \
        unsigned ad_rnd = base + 4u * ((threadIdx.x * 2654435761u >> 20) & 1023u); /* pseudo-random bins */ \
        unsigned ad_same = base + 4u * ((threadIdx.x >> 3) & 7u); /* 8 lanes per address */  \
        unsigned v0 = 1, v1 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0; \
        (void)ad_lin; (void)ad_l4; (void)ad_rnd; (void)ad_same;                              \
        for (int i = 0; i < ITER; i++) { BODY }                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (unsigned)lds[threadIdx.x]; \
    }
#define A8(x) x x x x x x x x
LDSKERNEL(k_ds_add_u32_lin, asm volatile(A8(A8("ds_add_u32 %0, %1\n")) "s_waitcnt lgkmcnt(0)\n" :: "v"(ad_l4), "v"(v0) : "memory");)
LDSKERNEL(k_ds_add_u32_rnd, asm volatile(A8(A8("ds_add_u32 %0, %1\n")) "s_waitcnt lgkmcnt(0)\n" :: "v"(ad_rnd), "v"(v0) : "memory");)
LDSKERNEL(k_ds_add_u32_same8, asm volatile(A8(A8("ds_add_u32 %0, %1\n")) "s_waitcnt lgkmcnt(0)\n" :: "v"(ad_same), "v"(v0) : "memory");)
LDSKERNEL(k_ds_add_u64_lin, asm volatile(A8(A8("ds_add_u64 %0, %1\n")) "s_waitcnt lgkmcnt(0)\n" :: "v"(ad_lin), "v"((unsigned long long)v0) : "memory");)
LDSKERNEL(k_ds_read_b32, asm volatile(A8("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(ad_l4) : "memory");)
LDSKERNEL(k_ds_read_b32_rnd, asm volatile(A8("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(ad_rnd) : "memory");)
LDSKERNEL(k_ds_read_u8_rnd, asm volatile(A8("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:256\n ds_read_u8 %2, %8 offset:512\n ds_read_u8 %3, %8 offset:768\n ds_read_u8 %4, %8 offset:1024\n ds_read_u8 %5, %8 offset:1280\n ds_read_u8 %6, %8 offset:1536\n ds_read_u8 %7, %8 offset:1792\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(ad_rnd) : "memory");)
LDSKERNEL(k_ds_write_b32, asm volatile(A8(A8("ds_write_b32 %0, %1\n")) "s_waitcnt lgkmcnt(0)\n" :: "v"(ad_l4), "v"(v0) : "memory");)
/* histogram-like mix: 1 ds_add per 2 VALU */
LDSKERNEL(k_mix_hist, asm volatile(A8(A8("v_bfe_u32 %2, %3, 8, 7\n v_lshl_add_u32 %2, %2, 2, %0\n ds_add_u32 %2, %1\n")) "s_waitcnt lgkmcnt(0)\n" : "+v"(ad_l4), "+v"(v0), "+v"(r0), "+v"(r1) :: "memory");)

struct Test { const char* name; void (*fn)(unsigned*, unsigned); double insts_per_iter; };

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, dev));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    unsigned* out;
    CHECK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Test tests[] = {
        {"v_add_u32", k_add, 64}, {"v_and_b32", k_and, 64}, {"v_xor_b32", k_xor, 64}, {"v_lshlrev_b32", k_lshl, 64},
        {"v_bitop3_b32", k_bitop3, 64}, {"v_lshl_or_b32", k_lshl_or, 64}, {"v_lshl_add_u32", k_lshl_add, 64},
        {"v_and_or_b32", k_and_or, 64}, {"v_or3_b32", k_or3, 64}, {"v_add3_u32", k_add3, 64}, {"v_xad_u32", k_xad, 64},
        {"v_bfe_u32", k_bfe, 64}, {"v_perm_b32", k_perm, 64}, {"v_alignbit_b32", k_alignbit, 64},
        {"v_alignbyte_b32", k_alignbyte, 64}, {"v_mad_u32_u24", k_mad_u24, 64}, {"v_mul_lo_u32", k_mul_lo, 64},
        {"v_mul_u32_u24", k_mul_u24, 64}, {"v_dot4_u32_u8", k_dot4, 64}, {"v_sad_u8", k_sad_u8, 64},
        {"v_bcnt_u32_b32", k_bcnt, 64}, {"v_min_u32", k_min, 64}, {"v_min3_u32", k_min3, 64},
        {"v_cndmask_b32", k_cndmask, 64}, {"v_cmp+v_cndmask (pairs)", k_cmp_cnd, 64}, {"v_add_co/addc", k_addc, 64},
        {"v_pk_add_u16", k_pk_add_u16, 64}, {"v_pk_lshlrev_b16", k_pk_lshl_b16, 64}, {"v_pk_sub_u16", k_pk_sub_u16, 64},
        {"v_lshlrev_b32_sdwa", k_sdwa_lshl, 64}, {"v_add_u32_sdwa", k_sdwa_add, 64}, {"v_add_u32_dpp row_shr", k_dpp_shr, 64},
        {"v_mov_b32_dpp row_shr", k_mov_dpp, 64}, {"v_readlane_b32", k_readlane, 64}, {"v_add_u32 (sgpr src)", k_add_sgpr, 64},
        {"SALU mix", k_salu, 64}, {"VALU+SALU interleaved (64 total)", k_mix_valu_salu, 64},
        {"v_add+v_perm interleaved", k_mix_add_perm, 64}, {"v_add_u32 dependent chain", k_dep_add, 64},
        {"ds_add_u32 linear", k_ds_add_u32_lin, 64}, {"ds_add_u32 pseudo-random", k_ds_add_u32_rnd, 64},
        {"ds_add_u32 8 lanes/address", k_ds_add_u32_same8, 64}, {"ds_add_u64 linear", k_ds_add_u64_lin, 64},
        {"ds_read_b32 linear", k_ds_read_b32, 64}, {"ds_read_b32 pseudo-random", k_ds_read_b32_rnd, 64},
        {"ds_read_u8 pseudo-random", k_ds_read_u8_rnd, 64}, {"ds_write_b32 linear", k_ds_write_b32, 64},
        {"hist mix (2 VALU + ds_add; counts 192)", k_mix_hist, 192},
    };
    const int occs[] = {1, 2, 4, 8}; /* waves per SIMD */
    printf("%-40s", "instruction");
    for (int o : occs) printf("  w/SIMD=%d", o);
    printf("   [wave-instructions per ns per CU; 2.4 = one per clock at 2.4 GHz]\n");
    for (const Test& t : tests) {
        printf("%-40s", t.name);
        for (int o : occs) {
            const int blocks = ncu * o; /* 256-thread block = 4 waves = one per SIMD */
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double winst = (double)blocks * 4 * ITER * t.insts_per_iter;
            printf("  %8.3f", winst / (best * 1e6) / ncu);
        }
        printf("\n");
    }
    return 0;
}

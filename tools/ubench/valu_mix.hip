// Profiling aid only: how mixed instruction streams issue on gfx950 (companion of valu_rates.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 512;
// %0..%7 chains, %8 = c (vgpr), %9 = d (vgpr), %10 = sgpr, %11 = lds address (vgpr), %12 lds data
#define KERNEL(NAME, BODY)                                                                    \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned c0) {                  \
        __shared__ unsigned lds[4096];                                                         \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;                              \
        __syncthreads();                                                                      \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned c = c0 | 1u, d = c0 + 3u, la = (unsigned)(size_t)lds + 4u * threadIdx.x, ld = 1;   \
        for (int i = 0; i < ITER; i++) {                                                      \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(c), "v"(d), "s"(c0), "v"(la), "v"(ld)                            \
                         : "vcc", "scc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); \
        }                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ lds[threadIdx.x];  \
    }
#define X8(s) s s s s s s s s
#define ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define DSADD "ds_add_u32 %11, %12\n"
#define DSREAD(i) "ds_read_b32 %" #i ", %11\n"
#define WAITL "s_waitcnt lgkmcnt(0)\n"

KERNEL(m_add_perm_alt, X8(ADD(0) PERM(1) ADD(2) PERM(3) ADD(4) PERM(5) ADD(6) PERM(7)))
KERNEL(m_add_perm_blk4, X8(ADD(0) ADD(1) ADD(2) ADD(3) PERM(4) PERM(5) PERM(6) PERM(7)))
KERNEL(m_add_perm_blk32, X8(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7)) X8(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7)) X8(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7)) X8(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7)) \
        X8(PERM(0) PERM(1) PERM(2) PERM(3) PERM(4) PERM(5) PERM(6) PERM(7)) X8(PERM(0) PERM(1) PERM(2) PERM(3) PERM(4) PERM(5) PERM(6) PERM(7)) X8(PERM(0) PERM(1) PERM(2) PERM(3) PERM(4) PERM(5) PERM(6) PERM(7)) X8(PERM(0) PERM(1) PERM(2) PERM(3) PERM(4) PERM(5) PERM(6) PERM(7)))
KERNEL(m_add7_perm1, X8(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) PERM(7)))
KERNEL(m_add3_perm1, X8(ADD(0) ADD(1) ADD(2) PERM(3) ADD(4) ADD(5) ADD(6) PERM(7)))
KERNEL(m_and_bitop_alt, X8(AND(0) BITOP(1) AND(2) BITOP(3) AND(4) BITOP(5) AND(6) BITOP(7)))
KERNEL(m_add_lshl_alt, X8(ADD(0) LSHL(1) ADD(2) LSHL(3) ADD(4) LSHL(5) ADD(6) LSHL(7)))
KERNEL(m_bitop_bfe_alt, X8(BITOP(0) BFE(1) BITOP(2) BFE(3) BITOP(4) BFE(5) BITOP(6) BFE(7)))
KERNEL(m_add_dsadd_1to1, X8(ADD(0) DSADD ADD(1) DSADD ADD(2) DSADD ADD(3) DSADD) WAITL)
KERNEL(m_add3_dsadd1, X8(ADD(0) ADD(1) ADD(2) DSADD ADD(4) ADD(5) ADD(6) DSADD) WAITL)
KERNEL(m_perm3_dsadd1, X8(PERM(0) PERM(1) PERM(2) DSADD PERM(4) PERM(5) PERM(6) DSADD) WAITL)
KERNEL(m_perm7_dsadd1, X8(PERM(0) PERM(1) PERM(2) PERM(3) PERM(4) PERM(5) PERM(6) DSADD) WAITL)
KERNEL(m_add_and_lit, X8("v_and_b32 %0, 0x7f7f7f7f, %0\n v_and_b32 %1, 0x7f7f7f7f, %1\n v_and_b32 %2, 0x7f7f7f7f, %2\n v_and_b32 %3, 0x7f7f7f7f, %3\n v_and_b32 %4, 0x7f7f7f7f, %4\n v_and_b32 %5, 0x7f7f7f7f, %5\n v_and_b32 %6, 0x7f7f7f7f, %6\n v_and_b32 %7, 0x7f7f7f7f, %7\n"))
KERNEL(m_and_inline, X8("v_and_b32 %0, 15, %0\n v_and_b32 %1, 15, %1\n v_and_b32 %2, 15, %2\n v_and_b32 %3, 15, %3\n v_and_b32 %4, 15, %4\n v_and_b32 %5, 15, %5\n v_and_b32 %6, 15, %6\n v_and_b32 %7, 15, %7\n"))
KERNEL(m_add_inline, X8("v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3\n v_add_u32 %4, 1, %4\n v_add_u32 %5, 1, %5\n v_add_u32 %6, 1, %6\n v_add_u32 %7, 1, %7\n"))
KERNEL(m_and_sgpr, X8("v_and_b32 %0, %10, %0\n v_and_b32 %1, %10, %1\n v_and_b32 %2, %10, %2\n v_and_b32 %3, %10, %3\n v_and_b32 %4, %10, %4\n v_and_b32 %5, %10, %5\n v_and_b32 %6, %10, %6\n v_and_b32 %7, %10, %7\n"))
KERNEL(m_or, X8("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n"))
KERNEL(m_sub, X8("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8\n"))
KERNEL(m_mov, X8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n"))
KERNEL(m_not, X8("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3\n v_not_b32 %4, %4\n v_not_b32 %5, %5\n v_not_b32 %6, %6\n v_not_b32 %7, %7\n"))
KERNEL(m_lshr, X8("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n"))
KERNEL(m_max, X8("v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8\n"))
KERNEL(m_cmp, X8("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n"))
KERNEL(m_and_e64, X8("v_and_b32_e64 %0, %0, %8\n v_and_b32_e64 %1, %1, %8\n v_and_b32_e64 %2, %2, %8\n v_and_b32_e64 %3, %3, %8\n v_and_b32_e64 %4, %4, %8\n v_and_b32_e64 %5, %5, %8\n v_and_b32_e64 %6, %6, %8\n v_and_b32_e64 %7, %7, %8\n"))
KERNEL(m_bitop_sgpr, X8("v_bitop3_b32 %0, %0, %10, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %10, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %10, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %10, %9 bitop3:0x96\n v_bitop3_b32 %4, %4, %10, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %10, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %10, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %10, %9 bitop3:0x96\n"))
KERNEL(m_add_salu_3to1, X8(ADD(0) ADD(1) ADD(2) "s_add_u32 s20, s20, %10\n" ADD(4) ADD(5) ADD(6) "s_and_b32 s21, s21, %10\n"))
KERNEL(m_perm_salu_3to1, X8(PERM(0) PERM(1) PERM(2) "s_add_u32 s20, s20, %10\n" PERM(4) PERM(5) PERM(6) "s_and_b32 s21, s21, %10\n"))
KERNEL(m_perm_salu_1to1, X8(PERM(0) "s_add_u32 s20, s20, %10\n" PERM(1) "s_and_b32 s21, s21, %10\n" PERM(2) "s_add_u32 s22, s22, %10\n" PERM(3) "s_and_b32 s23, s23, %10\n"))
KERNEL(m_perm_dsread_3to1, X8(PERM(0) PERM(1) PERM(2) DSREAD(3) PERM(4) PERM(5) PERM(6) DSREAD(7)) WAITL)

struct Test { const char* name; void (*fn)(unsigned*, unsigned); double n; };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    unsigned* out;
    CHECK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Test tests[] = {
        {"add,perm alternating", m_add_perm_alt, 64}, {"add x4, perm x4", m_add_perm_blk4, 64}, {"add x256, perm x256", m_add_perm_blk32, 512},
        {"add x7, perm x1", m_add7_perm1, 64}, {"add x3, perm x1", m_add3_perm1, 64}, {"and,bitop3 alternating", m_and_bitop_alt, 64},
        {"add,lshl alternating", m_add_lshl_alt, 64}, {"bitop3,bfe alternating", m_bitop_bfe_alt, 64},
        {"add,ds_add 1:1 (64 total)", m_add_dsadd_1to1, 64}, {"add x3, ds_add x1", m_add3_dsadd1, 64},
        {"perm x3, ds_add x1", m_perm3_dsadd1, 64}, {"perm x7, ds_add x1", m_perm7_dsadd1, 64},
        {"v_and_b32 literal", m_add_and_lit, 64}, {"v_and_b32 inline const", m_and_inline, 64}, {"v_add_u32 inline const", m_add_inline, 64},
        {"v_and_b32 sgpr", m_and_sgpr, 64}, {"v_or_b32", m_or, 64}, {"v_sub_u32", m_sub, 64}, {"v_mov_b32", m_mov, 64}, {"v_not_b32", m_not, 64},
        {"v_lshrrev_b32", m_lshr, 64}, {"v_max_u32", m_max, 64}, {"v_cmp_lt_u32", m_cmp, 64}, {"v_and_b32_e64", m_and_e64, 64},
        {"v_bitop3 sgpr src", m_bitop_sgpr, 64}, 
        {"add x3, salu x1", m_add_salu_3to1, 64}, {"perm x3, salu x1", m_perm_salu_3to1, 64}, {"perm,salu 1:1", m_perm_salu_1to1, 64},
        {"perm x3, ds_read x1", m_perm_dsread_3to1, 64},
    };
    const int occs[] = {1, 2, 4, 5, 8};
    printf("%-32s", "stream");
    for (int o : occs) printf("  w/SIMD=%d", o);
    printf("   [instructions (all kinds) per ns per CU]\n");
    for (const Test& t : tests) {
        printf("%-32s", t.name);
        for (int o : occs) {
            const int blocks = ncu * o;
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
            printf("  %8.3f", (double)blocks * 4 * ITER * t.n / (best * 1e6) / ncu);
        }
        printf("\n");
    }
    return 0;
}

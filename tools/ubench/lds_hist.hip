// Profiling aid only (not part of the product): what an LDS histogram update costs on gfx950 when lanes of a wave
// really collide -- ds_add_u32 into tables of 1024 / 2048 / 4096 bins with addresses drawn at random (duplicates as
// they come: 64 lanes into 1024 bins share an address twice per instruction on average), against the collision-free
// patterns of valu_rates.hip; ds_add_u64 in k_stats_sorted's cell layout (class row 8192 bytes apart, lane-major);
// rate = wave-instructions / (ns * CU).
//   hipcc --offload-arch=gfx950 -O3 -o lds_hist lds_hist.hip && ./lds_hist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 512;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: u32, random bins (BINS); 1: u32, random bins but lane-unique (a permutation: bank conflicts only);
// 2: u64 cells, class rows (random class of 4 per lane and slot), lane-major; 3: u64 linear; 4: u32 linear
// 5: u32 random bins, two updates folded: half the instructions per "base" is the caller's arithmetic, same kernel as 0
template <int MODE, int BINS>
__global__ void __launch_bounds__(1024) k_lds(unsigned* out, unsigned seed) {
    __shared__ unsigned long long lds[8192]; // 64 KB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;
    unsigned ad[8];
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned h = mix(seed + 977u * threadIdx.x + 131071u * j + 7u * blockIdx.x);
        if (MODE == 0) ad[j] = base + 4u * (h % BINS);
        if (MODE == 1) ad[j] = base + 4u * (((lane * 37u + (h & ~63u) * 64u) % BINS)); // 37 is odd: lane-unique mod 64.. BINS
        if (MODE == 2) ad[j] = base + (h & 3u) * 8192u + 8u * lane + 512u * j;
        if (MODE == 3) ad[j] = base + 8u * threadIdx.x % 2048u + 2048u * j;
        if (MODE == 4) ad[j] = base + 4u * threadIdx.x % 1024u + 1024u * j;
    }
    unsigned v0 = 1;
    unsigned long long v1 = 1;
    for (int i = 0; i < ITER; i++) {
        if (MODE == 2 || MODE == 3) {
            asm volatile("ds_add_u64 %0, %8\n ds_add_u64 %1, %8\n ds_add_u64 %2, %8\n ds_add_u64 %3, %8\n"
                         "ds_add_u64 %4, %8\n ds_add_u64 %5, %8\n ds_add_u64 %6, %8\n ds_add_u64 %7, %8\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(v1) : "memory");
        } else {
            asm volatile("ds_add_u32 %0, %8\n ds_add_u32 %1, %8\n ds_add_u32 %2, %8\n ds_add_u32 %3, %8\n"
                         "ds_add_u32 %4, %8\n ds_add_u32 %5, %8\n ds_add_u32 %6, %8\n ds_add_u32 %7, %8\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(v0) : "memory");
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)lds[threadIdx.x];
}

// the row of k_stats_sorted as an instruction stream: 8 x (ds_add_u64 cell + ds_add_u32 5-mer), or 8 x ds_add_u64 + 4 x ds_add_u32
// into a 4096-bin table (the 6-mer form), with NV filler VALU instructions per LDS pair
template <int KBINS, int KUPD, int NV>
__global__ void __launch_bounds__(1024) k_row(unsigned* out, unsigned seed) {
    __shared__ unsigned long long lds[10240]; // 64 KB cells + 16 KB (5-mers use 4 KB of it, 6-mers all)
    for (int i = threadIdx.x; i < 10240; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;
    const unsigned lane = threadIdx.x & 63u;
    unsigned ad[8], kd[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned h = mix(seed + 977u * threadIdx.x + 131071u * j + 7u * blockIdx.x);
        ad[j] = base + (h & 3u) * 8192u + 8u * lane + 512u * j;
        kd[j] = base + 65536u + 4u * ((h >> 8) % KBINS);
    }
    unsigned v0 = 1, f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3;
    unsigned long long v1 = 1;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            asm volatile("ds_add_u64 %0, %1\n" :: "v"(ad[j]), "v"(v1) : "memory");
            if (KUPD == 8 || (j & 1)) asm volatile("ds_add_u32 %0, %1\n" :: "v"(kd[j]), "v"(v0) : "memory");
#pragma unroll
            for (int q = 0; q < NV; q++) {
                if ((q & 3) == 0) asm volatile("v_bfe_u32 %0, %0, 1, 31\n" : "+v"(f0));
                if ((q & 3) == 1) asm volatile("v_add_u32 %0, %0, %1\n" : "+v"(f1) : "v"(f0));
                if ((q & 3) == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n" : "+v"(f2) : "v"(f1));
                if ((q & 3) == 3) asm volatile("v_and_b32 %0, %0, %1\n" : "+v"(f3) : "v"(f2));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n" ::: "memory");
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)lds[threadIdx.x] ^ f0 ^ f1 ^ f2 ^ f3;
}

struct Test { const char* name; void (*fn)(unsigned*, unsigned); double insts_per_iter; };

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    unsigned* out;
    CHECK(hipMalloc(&out, (size_t)ncu * 8 * 1024 * 4 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Test tests[] = {
        {"ds_add_u32 linear", k_lds<4, 1024>, 8},
        {"ds_add_u32 lane-unique pseudo-random bins", k_lds<1, 1024>, 8},
        {"ds_add_u32 random, 256 bins (dups ~8/inst)", k_lds<0, 256>, 8},
        {"ds_add_u32 random, 1024 bins (dups ~2/inst)", k_lds<0, 1024>, 8},
        {"ds_add_u32 random, 2048 bins", k_lds<0, 2048>, 8},
        {"ds_add_u32 random, 4096 bins (dups ~0.5/inst)", k_lds<0, 4096>, 8},
        {"ds_add_u32 random, 16384 bins", k_lds<0, 16384>, 8},
        {"ds_add_u64 linear", k_lds<3, 1>, 8},
        {"ds_add_u64 class rows, lane-major", k_lds<2, 1>, 8},
        {"row: 8 u64 + 8 u32/1024 bins (LDS ops)", k_row<1024, 8, 0>, 16},
        {"row: 8 u64 + 4 u32/4096 bins (LDS ops)", k_row<4096, 4, 0>, 12},
        {"row: 8 u64 + 8 u32/1024 + 5 VALU each (rows*16)", k_row<1024, 8, 5>, 16},
        {"row: 8 u64 + 4 u32/4096 + 5 VALU each (rows*16)", k_row<4096, 4, 5>, 16},
        {"row: 8 u64 + 8 u32/1024 + 10 VALU each (rows*16)", k_row<1024, 8, 10>, 16},
        {"row: 8 u64 + 4 u32/4096 + 10 VALU each (rows*16)", k_row<4096, 4, 10>, 16},
    };
    const int occs[] = {1, 2, 4, 8};
    printf("%-52s", "instruction stream");
    for (int o : occs) printf("  w/SIMD=%d", o);
    printf("   [per ns per CU; 2.4 = one per clock at 2.4 GHz]\n");
    for (const Test& t : tests) {
        printf("%-52s", t.name);
        for (int o : occs) {
            /* 64 KB+ of LDS per block: two blocks per CU at most -- 1, 2 waves per SIMD as 256-thread blocks, 4, 8 as 1024-thread blocks */
            const int threads = o <= 2 ? 256 : 1024;
            const int blocks = ncu * (o <= 2 ? o : o / 4);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u + r);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double winst = (double)blocks * (threads / 64) * ITER * t.insts_per_iter;
            printf("  %8.3f", winst / (best * 1e6) / ncu);
        }
        printf("\n");
    }
    return 0;
}

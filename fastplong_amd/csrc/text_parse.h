/*
 * text_parse.h -- FASTQ text -> CSR batch ON THE DEVICE (fpl_process_text_async, include/fastplong_amd.h).
 *
 * The reference's reader builds Read objects line by line on one thread (FastqReader::getLine / ::read,
 * src/fastqreader.cpp:219-347); this host used to locate the lines on its CPUs and copy every base and quality into
 * page-locked CSR arrays -- the copy was its ceiling.  Here the host only uploads the chunk's bytes as they lie in the file
 * (2.02 bytes per base over the link, as before) and these kernels do the reader's work where the bandwidth is:
 *
 *   k_text_count    '\n' per 4 KiB block, and the bytes that make a chunk IRREGULAR at first sight (a '\r' that is not
 *                   followed by '\n': the reference ends a line there, src/fastqreader.cpp:227-246)
 *   k_text_scan     exclusive prefix sums of the block counts, the number of lines
 *   k_text_fill     the position of every '\n', in order
 *   k_text_records  lane = record (four lines): where its lines start, "\r\n" taken off, the checks of FastqReader::read
 *                   ('@' in front of the name, '+' in front of the third line, as many qualities as bases,
 *                   src/fastqreader.cpp:312-341) -- a chunk that fails one is IRREGULAR as a whole
 *   k_text_offsets  exclusive prefix sums of the read lengths: the CSR offsets, the longest read, the number of bases
 *   k_text_gather   a wave per record: its bases and qualities to where the per-read kernels expect them
 *
 * REGULAR text is what every FASTQ writer produces: records of exactly four lines, every line ended by "\n" or "\r\n", the
 * last one too.  Anything else -- blank lines, headers that do not start with '@' (the reference skips lines until one does),
 * a lone '\r', a missing final line break, a malformed record -- is reported as FPL_TEXT_IRREGULAR with nothing processed, and
 * the caller parses that chunk with the host's reader, which reproduces the reference's behaviour byte for byte.
 */
#ifndef FPL_TEXT_PARSE_H
#define FPL_TEXT_PARSE_H

#include "dev_prims.h"

namespace fpl {

constexpr int TP_BLOCK_BYTES = 4096; /* per block of 256 threads: 16 bytes a thread */
constexpr int TP_THREADS = 256;

/* the header the host reads back once per chunk, between the parse and the per-read kernels */
struct TextHeader {
    u32 n_lines;     /* '\n' bytes of the chunk */
    u32 n_records;   /* n_lines / 4 */
    u32 status;      /* bit 0: irregular text (see above), bit 1: more records than the caller's buffers hold */
    u32 max_len;     /* longest read */
    u64 n_bases;     /* bases of all reads = CSR offset of the end */
    u64 bad_record;  /* lowest record that failed a check (~0: none) */
};

__device__ __forceinline__ u32 tp_eq_mask16(const u32 (&w)[4], u32 byte) {
    /* bit i: byte i of the 16 equals `byte` */
    u32 m = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const u32 x = w[d] ^ (byte * 0x01010101u);
        /* bit 7 of every zero byte of x (exact: no borrow crosses a byte whose own test could be wrong) */
        const u32 z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
        m |= (((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu) << (4 * d);
    }
    return m;
}

/* this thread's 16 bytes (zeros behind the end of the text) */
__device__ __forceinline__ void tp_load16(const u8* __restrict__ text, u64 n, u64 at, u32 (&w)[4]) {
    if (at + 16 <= n) {
        const u32x4 v = *(const u32x4*)(text + at); /* (text is 16-byte aligned: hipMalloc) */
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            u32 x = 0;
            for (int b = 0; b < 4; b++) {
                const u64 p = at + 4 * d + b;
                if (p < n) x |= (u32)text[p] << (8 * b);
            }
            w[d] = x;
        }
    }
}

__global__ void __launch_bounds__(TP_THREADS)
k_text_count(const u8* __restrict__ text, u64 n, u32* __restrict__ blkcnt, TextHeader* __restrict__ hdr) {
    __shared__ u32 wsum[TP_THREADS / 64];
    const u64 at = (u64)blockIdx.x * TP_BLOCK_BYTES + 16ull * threadIdx.x;
    u32 w[4] = {0, 0, 0, 0};
    u32 cnt = 0;
    bool odd = false;
    if (at < n) {
        tp_load16(text, n, at, w);
        const u32 nl = tp_eq_mask16(w, '\n'), cr = tp_eq_mask16(w, '\r');
        cnt = (u32)__popc(nl);
        /* every '\r' must have a '\n' behind it (byte 16 of this thread is the next thread's byte 0) */
        u32 next_nl = nl >> 1;
        if (at + 16 < n && text[at + 16] == '\n') next_nl |= 1u << 15;
        odd = (cr & ~next_nl) != 0;
    }
    const u32 ws = wave_sum_u32(cnt);
    if (lane_id() == 0) wsum[wave_in_block()] = ws;
    if (wave_ballot(odd) && lane_id() == 0) atomicOr(&hdr->status, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int i = 0; i < TP_THREADS / 64; i++) t += wsum[i];
        blkcnt[blockIdx.x] = t;
    }
}

/* one block: blkcnt[i] -> the sum of the counts in front of block i; hdr->n_lines */
__global__ void __launch_bounds__(1024)
k_text_scan(u32* __restrict__ blkcnt, u32 nblk, TextHeader* __restrict__ hdr) {
    __shared__ u32 wsum[16];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < nblk; base += 1024) { /* block-uniform */
        const u32 i = base + threadIdx.x;
        const u32 v = i < nblk ? blkcnt[i] : 0u;
        const u32 incl = wave_scan_incl_u32(v);
        if (lane_id() == 63) wsum[wave_in_block()] = incl;
        __syncthreads();
        u32 run = carry + incl - v;
        for (int k = 0; k < wave_in_block(); k++) run += wsum[k];
        if (i < nblk) blkcnt[i] = run;
        __syncthreads();
        if (threadIdx.x == 1023) carry = run + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) hdr->n_lines = carry;
}

__global__ void __launch_bounds__(TP_THREADS)
k_text_fill(const u8* __restrict__ text, u64 n, const u32* __restrict__ blkoff, u32* __restrict__ nl_pos, u32 nl_cap) {
    __shared__ u32 wsum[TP_THREADS / 64];
    const u64 at = (u64)blockIdx.x * TP_BLOCK_BYTES + 16ull * threadIdx.x;
    u32 w[4] = {0, 0, 0, 0};
    u32 nl = 0;
    if (at < n) {
        tp_load16(text, n, at, w);
        nl = tp_eq_mask16(w, '\n');
    }
    const u32 cnt = (u32)__popc(nl);
    const u32 incl = wave_scan_incl_u32(cnt);
    if (lane_id() == 63) wsum[wave_in_block()] = incl;
    __syncthreads();
    u32 k = blkoff[blockIdx.x] + incl - cnt;
    for (int i = 0; i < wave_in_block(); i++) k += wsum[i];
    while (nl) {
        const int b = __ffs((int)nl) - 1;
        nl &= nl - 1;
        if (k < nl_cap) nl_pos[k] = (u32)(at + (u64)b);
        k++;
    }
}

/* lane = record.  line[4 r + j] = where line j of record r starts; len[r] = its bases */
__global__ void __launch_bounds__(256)
k_text_records(const u8* __restrict__ text, u64 n, const u32* __restrict__ nl_pos, u32 rec_cap, TextHeader* __restrict__ hdr,
               u32* __restrict__ line, u32* __restrict__ len) {
    const u32 n_lines = hdr->n_lines;
    const u32 n_rec = n_lines / 4;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr->n_records = n_rec;
        u32 st = 0;
        if (n_lines % 4 != 0 || (n > 0 && text[n - 1] != '\n')) st |= 1u; /* a record cut short / no line break at the end */
        if (n_rec > rec_cap) st |= 2u;
        if (st) atomicOr(&hdr->status, st);
    }
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rec && r < rec_cap; r += gridDim.x * blockDim.x) {
        u32 L[5];
        L[0] = r ? nl_pos[4 * r - 1] + 1u : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) L[j + 1] = nl_pos[4 * r + j] + 1u;
        u32 ll[4]; /* line lengths without the line break */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 e = L[j + 1] - 1u; /* the '\n' */
            if (e > L[j] && text[e - 1] == '\r') e--;
            ll[j] = e - L[j];
            line[4 * (size_t)r + j] = L[j];
        }
        const bool good = ll[0] > 0 && text[L[0]] == '@' && ll[2] > 0 && text[L[2]] == '+' && ll[1] == ll[3];
        len[r] = ll[1];
        if (!good) {
            atomicOr(&hdr->status, 1u);
            atomicMin((unsigned long long*)&hdr->bad_record, (unsigned long long)r);
        }
    }
}

/* one block: len[] -> off[] (n_rec + 1 entries), the longest read, the number of bases */
__global__ void __launch_bounds__(1024)
k_text_offsets(const u32* __restrict__ len, u32 rec_cap, TextHeader* __restrict__ hdr, uint64_t* __restrict__ off) {
    __shared__ u64 wsum[16];
    __shared__ u64 carry;
    __shared__ u32 wmax[16];
    const u32 n_rec = min(hdr->n_records, rec_cap);
    if (threadIdx.x == 0) carry = 0;
    u32 mx = 0;
    __syncthreads();
    for (u32 base = 0; base < n_rec; base += 1024) { /* block-uniform */
        const u32 i = base + threadIdx.x;
        const u32 v = i < n_rec ? len[i] : 0u;
        mx = max(mx, v);
        /* (64-bit sums as two 32-bit scans would need a carry: the values are below 2^32 and a block's sum below 2^42 -- scan the
           halves of a 64-bit value through the 32-bit primitive with the low half's overflow counted) */
        u64 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 up = shfl_u64(incl, lane_id() - d < 0 ? lane_id() : lane_id() - d);
            if (lane_id() >= d) incl += up;
        }
        if (lane_id() == 63) wsum[wave_in_block()] = incl;
        __syncthreads();
        u64 run = carry + incl - v;
        for (int k = 0; k < wave_in_block(); k++) run += wsum[k];
        if (i < n_rec) off[i] = run;
        __syncthreads();
        if (threadIdx.x == 1023) carry = run + v;
        __syncthreads();
    }
    const u32 wm = wave_max_u32(mx);
    if (lane_id() == 0) wmax[wave_in_block()] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = 0;
        for (int k = 0; k < 16; k++) m = max(m, wmax[k]);
        hdr->max_len = m;
        hdr->n_bases = carry;
        off[n_rec] = carry;
    }
}

/* a wave per record: bases and qualities out of the text into the CSR arrays (16 bytes a lane and step; neither side is aligned) */
__global__ void __launch_bounds__(256)
k_text_gather(const u8* __restrict__ text, const u32* __restrict__ line, const u32* __restrict__ len,
              const uint64_t* __restrict__ off, const TextHeader* __restrict__ hdr, u32 rec_cap, u8* __restrict__ seq,
              u8* __restrict__ qual) {
    if (hdr->status) return; /* (nothing of an irregular chunk is used) */
    const u32 n_rec = min(hdr->n_records, rec_cap);
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const u32 lane = (u32)lane_id();
    for (u32 r = wave; r < n_rec; r += n_waves) { /* wave-uniform */
        const u32 l = len[r];
        const uint64_t o = off[r];
#pragma unroll
        for (int which = 0; which < 2; which++) {
            const u8* src = text + line[4 * (size_t)r + (which ? 3 : 1)];
            u8* dst = (which ? qual : seq) + o;
            const u32 whole = l & ~15u;
            for (u32 i = 16 * lane; i < whole; i += 16 * 64) {
                u32x4 v;
                __builtin_memcpy(&v, src + i, 16);
                __builtin_memcpy(dst + i, &v, 16);
            }
            if (lane < (l & 15u)) dst[whole + lane] = src[whole + lane];
        }
    }
}

}  // namespace fpl
#endif

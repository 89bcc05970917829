/*
 * fpl_oracle.c -- TEST INFRASTRUCTURE ONLY (see fpl_oracle.h for the parity status).
 *
 * Plain-C restatement of the per-read hot path of OpenGene/fastplong v0.4.1.  Each function
 * cites the reference file:line it follows; loop bounds, tie-breaks and silent drops are the
 * reference's, including its off-by-ones.  The reference mutates std::string in place; here
 * a read is a window [start, start+len) on the immutable original bytes.
 */
#include "fpl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------------------------------
 * edit_distance -- reference src/editdistance.cpp:100-126.  The reference computes the exact
 * global Levenshtein distance with the Myers/Hyyro bit-parallel recurrence (:30-61, blocks of
 * 64 columns, :87-98) and falls back to the textbook DP (:65-76) beyond 640 columns; asize==0
 * returns bsize and vice versa (:101-102).  Restated as the textbook two-row DP.
 * ---------------------------------------------------------------------------------------- */
unsigned orc_edit_distance(const char* a, unsigned asize, const char* b, unsigned bsize) {
    if (asize == 0) return bsize;
    if (bsize == 0) return asize;
    unsigned* prev = (unsigned*)malloc(sizeof(unsigned) * (bsize + 1) * 2);
    unsigned* cur = prev + (bsize + 1);
    for (unsigned j = 0; j <= bsize; j++) prev[j] = j;
    for (unsigned i = 1; i <= asize; i++) {
        cur[0] = i;
        for (unsigned j = 1; j <= bsize; j++) {
            unsigned d = ORC_MIN(prev[j], cur[j - 1]) + 1;
            unsigned s = prev[j - 1] + (a[i - 1] == b[j - 1] ? 0 : 1);
            cur[j] = ORC_MIN(d, s);
        }
        unsigned* t = prev;
        prev = cur;
        cur = t;
    }
    unsigned res = prev[bsize];
    free(prev < cur ? prev : cur);
    return res;
}

static int hamming(const char* r, const char* a, int alen) {
    /* the Highway loop of src/adaptertrimmer.cpp:90-97 counts lanes where the bytes differ */
    int mm = 0;
    for (int i = 0; i < alen; i++)
        if (r[i] != a[i]) mm++;
    return mm;
}

/* ------------------------------------------------------------------------------------------
 * AdapterTrimmer::searchAdapter -- reference src/adaptertrimmer.cpp:59-166.
 * ---------------------------------------------------------------------------------------- */
int orc_search_adapter(const char* rdata, int rlen, const char* adata, int alen, double edMax,
                       int searchStart, int searchLen, int asLeft, int asRight) {
    int minMismatch = 99999; /* :65 */
    int pos = -1;
    int threshold = (int)round(edMax * alen); /* :73 */
    int searchEnd = rlen;
    if (searchLen > 0) searchEnd = ORC_MIN(rlen, searchLen + searchStart); /* :76-79 */
    if (searchStart + alen > rlen) return -1; /* :81-82 */

    if (asLeft) { /* :84-107: leftmost hit wins at once, no edit-distance check */
        for (int p = searchStart; p < searchEnd - alen; p++) {
            int mismatch = hamming(rdata + p, adata, alen);
            if (mismatch <= threshold) return p;
            if (mismatch <= minMismatch) {
                minMismatch = mismatch;
                pos = p;
            }
        }
    } else if (asRight && searchEnd > alen) { /* :109-131: rightmost hit wins at once */
        for (int p = searchEnd - alen; p >= searchStart; p--) {
            int mismatch = hamming(rdata + p, adata, alen);
            if (mismatch <= threshold) return p;
            if (mismatch <= minMismatch) {
                minMismatch = mismatch;
                pos = p;
            }
        }
    } else { /* :133-151: first global minimum */
        for (int p = searchStart; p < searchEnd - alen; p++) {
            int mismatch = hamming(rdata + p, adata, alen);
            if (mismatch < minMismatch) {
                minMismatch = mismatch;
                pos = p;
            }
        }
    }
    if (pos >= 0) { /* :154-162 */
        int ed = (int)orc_edit_distance(rdata + pos, alen, adata, alen);
        return ed <= threshold ? pos : -1;
    }
    return -1;
}

/* Read::trimFront -- reference src/read.cpp:69-73.  A negative len reaches
 * std::string::erase(0, (size_t)len) and erases everything. */
static void read_trim_front(orc_read* r, int len) {
    len = ORC_MIN(r->len - 1, len);
    if (len < 0) {
        r->start += r->len;
        r->len = 0;
        return;
    }
    r->start += len;
    r->len -= len;
}

/* Read::resize -- reference src/read.cpp:62-67 */
static void read_resize(orc_read* r, int len) {
    if (len > r->len || len < 0) return;
    r->len = len;
}

/* ------------------------------------------------------------------------------------------
 * AdapterTrimmer::trimBySequenceStart -- reference src/adaptertrimmer.cpp:168-236
 * ---------------------------------------------------------------------------------------- */
int orc_trim_start(orc_read* r, const char* adata, int alen, double edMax, int ext, int* key_len) {
    const int WINDOW = 200, PATTERN_LEN = 16;
    int rlen = r->len;
    const char* rdata = r->seq + r->start;
    *key_len = 0;
    if (rlen < PATTERN_LEN) return 0; /* :178-179 */
    int plen = ORC_MIN(PATTERN_LEN, alen);

    int mpos = orc_search_adapter(rdata, rlen, adata, alen, edMax, 0, WINDOW, 0, 1); /* :184 */
    if (mpos >= 0) {
        mpos = ORC_MIN(mpos + ext, rlen - alen); /* :187 */
        *key_len = alen;                         /* :188-189 addAdapterTrimmed(adapterseq) */
        read_trim_front(r, mpos + alen);
        return mpos + alen;
    }
    int mined = -1, pos = -1;
    for (int p = 0; p < rlen - plen && p < WINDOW - plen; p++) { /* :202-216 */
        int ed = (int)orc_edit_distance(rdata + p, plen, adata + alen - plen, plen);
        if (ed <= round(edMax * plen)) {
            if (pos < 0) {
                pos = p;
                mined = ed;
            } else if (ed >= mined) {
                /* keep the earlier one */
            } else {
                pos = p;
                mined = ed;
            }
        }
    }
    if (pos >= 0) { /* :218-233 */
        int cmplen = ORC_MIN(pos + plen, alen);
        int ed = (int)orc_edit_distance(rdata + pos + plen - cmplen, cmplen, adata + alen - cmplen, cmplen);
        if (ed <= round(edMax * cmplen)) {
            pos = ORC_MIN(pos + ext, rlen - alen);
            *key_len = cmplen; /* addAdapterTrimmed(adapterseq.substr(alen - cmplen, cmplen)) */
            read_trim_front(r, pos + plen);
            return pos + plen;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * AdapterTrimmer::trimBySequenceEnd -- reference src/adaptertrimmer.cpp:238-302
 * ---------------------------------------------------------------------------------------- */
int orc_trim_end(orc_read* r, const char* adata, int alen, double edMax, int ext, int* key_len) {
    const int WINDOW = 200, PATTERN_LEN = 16;
    int rlen = r->len;
    const char* rdata = r->seq + r->start;
    *key_len = 0;
    if (rlen < PATTERN_LEN) return 0; /* :248-249 (returns false) */
    int plen = ORC_MIN(PATTERN_LEN, alen);

    int searchStart = ORC_MAX(0, rlen - WINDOW);
    int mpos = orc_search_adapter(rdata, rlen, adata, alen, edMax, searchStart, WINDOW, 1, 0); /* :255 */
    if (mpos >= 0) {
        mpos = ORC_MAX(0, mpos - ext); /* :258 */
        *key_len = alen;
        read_resize(r, mpos);
        return rlen - mpos;
    }
    int mined = -1, pos = -1;
    for (int p = 0; p < rlen - plen && p < WINDOW - plen; p++) { /* :273-286 */
        int ed = (int)orc_edit_distance(rdata + rlen - plen - p, plen, adata, plen);
        if (ed <= round(edMax * plen)) {
            if (pos < 0) {
                pos = p;
                mined = ed;
            } else if (ed > mined) {
                break;
            } else {
                pos = p;
                mined = ed;
            }
        }
    }
    if (pos > 0) { /* :288 strict */
        int cmplen = ORC_MIN(pos + plen, alen);
        if ((int)orc_edit_distance(rdata + rlen - plen - pos, cmplen, adata, cmplen) <= round(edMax * cmplen)) {
            pos = ORC_MIN(pos + ext, rlen - plen);
            *key_len = cmplen; /* addAdapterTrimmed(adapterseq.substr(0, cmplen)) */
            read_resize(r, rlen - plen - pos);
            return pos + plen;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * AdapterTrimmer::findMiddleAdapters -- reference src/adaptertrimmer.cpp:13-40
 * ---------------------------------------------------------------------------------------- */
int orc_find_middle(const orc_read* r, const char* sa, int salen, const char* ea, int ealen,
                    double edMax, int ext, int* start, int* len) {
    const char* rdata = r->seq + r->start;
    int rlen = r->len;
    *len = -1;
    int sp = orc_search_adapter(rdata, rlen, sa, salen, edMax, 0, -1, 0, 0);
    int ep = orc_search_adapter(rdata, rlen, ea, ealen, edMax, 0, -1, 0, 0);
    if (sp >= 0 && ep >= 0) {
        int s = ORC_MIN(sp, ep);
        int e = ORC_MAX(sp + salen, ep + ealen);
        s = ORC_MAX(0, s - ext);
        e = ORC_MIN(rlen, e + ext);
        *start = s;
        *len = e - s;
        return 1;
    }
    if (sp >= 0) {
        int e = ORC_MIN(rlen, sp + salen + ext);
        *start = ORC_MAX(0, sp - ext);
        *len = e - *start;
        return 1;
    }
    if (ep >= 0) {
        int e = ORC_MIN(rlen, ep + ealen + ext);
        *start = ORC_MAX(0, ep - ext);
        *len = e - *start;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Filter::trimAndCut -- reference src/filter.cpp:130-232
 * ---------------------------------------------------------------------------------------- */
int orc_trim_and_cut(orc_read* r, const fpl_options* o, int* frontTrimmed) {
    int front = o->trim_front, tail = o->trim_tail;
    *frontTrimmed = 0;
    if (front == 0 && tail == 0 && !o->cut_front && !o->cut_tail) return 0; /* :133-134 */

    int rlen = r->len - front - tail; /* :137 */
    if (rlen < 0) return -1;
    if (front == 0 && !o->cut_front && !o->cut_tail) { /* :141-143 */
        read_resize(r, rlen);
        return 0;
    } else if (!o->cut_front && !o->cut_tail) { /* :144-151 */
        r->start += front;
        r->len = rlen;
        *frontTrimmed = front;
        return 0;
    }

    int l = r->len;
    const char* qualstr = r->qual + r->start;
    const char* seq = r->seq + r->start;
    if (o->cut_front) { /* :159-189 */
        int w = o->cut_front_window;
        int s = front;
        if (l - front - tail - w <= 0) return -1;
        int totalQual = 0;
        for (int i = 0; i < w - 1; i++) totalQual += qualstr[s + i];
        for (s = front; s + w < l - tail; s++) {
            totalQual += qualstr[s + w - 1];
            if (s > front) totalQual -= qualstr[s - 1];
            if ((double)totalQual / (double)w >= 33 + o->cut_front_quality) break;
        }
        if (s > 0) s = s + w - 1;
        while (s < l && seq[s] == 'N') s++;
        front = s;
        rlen = l - front - tail;
    }
    if (o->cut_tail) { /* :191-219 */
        int w = o->cut_tail_window;
        if (l - front - tail - w <= 0) return -1;
        int totalQual = 0;
        int t = l - tail - 1;
        for (int i = 0; i < w - 1; i++) totalQual += qualstr[t - i];
        for (t = l - tail - 1; t - w >= front; t--) {
            totalQual += qualstr[t - w + 1];
            if (t < l - tail - 1) totalQual -= qualstr[t + 1];
            if ((double)totalQual / (double)w >= 33 + o->cut_tail_quality) break;
        }
        if (t < l - 1) t = t - w + 1;
        while (t >= 0 && seq[t] == 'N') t--;
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return -1; /* :221-222 */
    r->start += front;
    r->len = rlen;
    *frontTrimmed = front;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * PolyX::trimPolyX -- reference src/polyx.cpp:11-78.  Index -1 (read when the whole read is
 * polyX, :71) is undefined behaviour in the reference; defined here as "not the poly base".
 * ---------------------------------------------------------------------------------------- */
int orc_trim_polyx(orc_read* r, int compareReq, int* poly_out, int* trimmed_out) {
    const int allowOneMismatchForEach = 8, maxMismatch = 5;
    static const char ATCG[4] = {'A', 'T', 'C', 'G'};
    const char* data = r->seq + r->start;
    int rlen = r->len;
    int n[4] = {0, 0, 0, 0};
    int pos;
    for (pos = 0; pos < rlen; pos++) {
        switch (data[rlen - pos - 1]) {
            case 'A': n[0]++; break;
            case 'T': n[1]++; break;
            case 'C': n[2]++; break;
            case 'G': n[3]++; break;
            case 'N': n[0]++; n[1]++; n[2]++; n[3]++; break;
            default: break;
        }
        int cmp = pos + 1;
        int allowed = ORC_MIN(maxMismatch, cmp / allowOneMismatchForEach);
        int needToBreak = 1;
        for (int b = 0; b < 4; b++)
            if (cmp - n[b] <= allowed) needToBreak = 0;
        if (needToBreak && (pos >= allowOneMismatchForEach || pos + 1 >= compareReq - 1)) break;
    }
    if (pos + 1 >= compareReq) { /* :57 */
        int poly = 0, maxCount = -1;
        for (int b = 0; b < 4; b++)
            if (n[b] > maxCount) {
                maxCount = n[b];
                poly = b;
            }
        char polyBase = ATCG[poly];
        while (pos >= 0) { /* :71 */
            int idx = rlen - pos - 1;
            if (idx >= 0 && data[idx] == polyBase) break;
            pos--;
        }
        read_resize(r, rlen - pos - 1);
        *poly_out = poly;
        *trimmed_out = pos + 1;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Filter::passFilter / passLowComplexityFilter -- reference src/filter.cpp:12-81
 * ---------------------------------------------------------------------------------------- */
int orc_pass_filter(const orc_read* r, const fpl_options* o) {
    if (r->len == 0) return FPL_FAIL_LENGTH;
    int rlen = r->len;
    int lowQualNum = 0, nBaseNum = 0, totalQual = 0;
    const char* seqstr = r->seq + r->start;
    const char* qualstr = r->qual + r->start;
    if (o->qual_filter || o->length_filter) {
        for (int i = 0; i < rlen; i++) {
            char base = seqstr[i];
            char qual = qualstr[i];
            totalQual += qual - 33;
            if (qual < (char)o->qualified_qual) lowQualNum++;
            if (base == 'N') nBaseNum++;
        }
    }
    if (o->qual_filter) {
        if (lowQualNum > (o->unqualified_percent_limit * rlen / 100.0)) return FPL_FAIL_QUALITY;
        else if (o->avg_qual_req > 0 && (totalQual / rlen) < o->avg_qual_req) return FPL_FAIL_QUALITY;
        else if (nBaseNum * 100 > rlen * o->n_base_percent_limit) return FPL_FAIL_N_BASE;
        else if (o->n_base_limit != 1000000 && nBaseNum > o->n_base_limit) return FPL_FAIL_N_BASE;
    }
    if (o->length_filter) {
        if (rlen < o->required_length) return FPL_FAIL_LENGTH;
        if (o->max_length > 0 && rlen > o->max_length) return FPL_FAIL_TOO_LONG;
    }
    if (o->complexity_filter) {
        int diff = 0;
        int ok;
        if (rlen <= 1) ok = 0;
        else {
            for (int i = 0; i < rlen - 1; i++)
                if (seqstr[i] != seqstr[i + 1]) diff++;
            /* src/main.cpp:219 stores min(100,max(0,Y))/100.0 */
            int y = ORC_MIN(100, ORC_MAX(0, o->complexity_percent));
            double threshold = y / 100.0;
            ok = ((double)diff / (double)(rlen - 1) >= threshold);
        }
        if (!ok) return FPL_FAIL_COMPLEXITY;
    }
    return FPL_PASS_FILTER;
}

/* Stats::base2val -- reference src/stats.cpp:411-425 */
static int base2val(char base) {
    switch (base) {
        case 'A': return 0;
        case 'T':
        case 'U': return 1;
        case 'C': return 2;
        case 'G': return 3;
        default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * Stats::statRead -- reference src/stats.cpp:265-375, written with the reference's rolling
 * k-mer state machine (needFullCompute) as is.  Quality bytes are assumed < 128 (the
 * reference indexes a long[128] with a signed char).
 * ---------------------------------------------------------------------------------------- */
void orc_stat_read(int64_t* st, uint32_t C, const orc_read* r, uint8_t* median_out) {
    int len = r->len;
    const char* seqstr = r->seq + r->start;
    const char* qualstr = r->qual + r->start;
    int64_t* bqh = st + FPL_ST_BASE_QUAL_HIST(C);
    int64_t* kmerv = st + FPL_ST_KMER(C);
    st[FPL_ST_LENGTH_SUM(C)] += len;
    int qualHist[128];
    memset(qualHist, 0, sizeof(qualHist));
    int kmer = 0;
    int needFullCompute = 1;
    for (int i = 0; i < len; i++) {
        char base = seqstr[i];
        char qual = qualstr[i];
        int b = base & 0x07;
        bqh[(int)qual]++;
        qualHist[(int)qual]++;
        if (qual >= '?') {
            st[FPL_ST_CYC(i, 3, b)]++;
            st[FPL_ST_CYC(i, 2, b)]++;
        } else if (qual >= '5') {
            st[FPL_ST_CYC(i, 2, b)]++;
        }
        st[FPL_ST_CYC(i, 0, b)]++;
        st[FPL_ST_CYC(i, 1, b)] += (qual - 33);
        if (base == 'N') {
            needFullCompute = 1;
            continue;
        }
        if (i < 4) continue;
        if (!needFullCompute) {
            int val = base2val(base);
            if (val < 0) {
                needFullCompute = 1;
                continue;
            } else {
                kmer = ((kmer << 2) & 0x3FC) | val;
                kmerv[kmer]++;
            }
        } else {
            int valid = 1;
            kmer = 0;
            for (int k = 0; k < 5; k++) {
                int val = base2val(seqstr[i - 4 + k]);
                if (val < 0) {
                    valid = 0;
                    break;
                }
                kmer = ((kmer << 2) & 0x3FC) | val;
            }
            if (!valid) {
                needFullCompute = 1;
                continue;
            } else {
                kmerv[kmer]++;
                needFullCompute = 0;
            }
        }
    }
    uint8_t med = 0;
    if (len > 0) { /* :352-363 */
        int total = 0;
        int median = 0;
        int half = len >> 1;
        while (1) {
            total += qualHist[median];
            if (total > half) break;
            median++;
        }
        st[FPL_ST_MEDIAN_HIST(C) + median]++;
        st[FPL_ST_MEDIAN_BASES(C) + median] += len;
        med = (uint8_t)median;
    }
    if (median_out) *median_out = med;
    st[FPL_ST_READS(C)]++;
}

/* ------------------------------------------------------------------------------------------
 * SingleEndProcessor::processSingleEnd, one read -- reference src/seprocessor.cpp:186-295
 * (--break / --mask, :234-262, are not part of this path yet).
 * ---------------------------------------------------------------------------------------- */
/* Filter::detectLowQualityRegions, src/filter.cpp:83-128, restated literally -- including the warm-up loop
 * `for (i = start; i < windowSize - 1 && i < l; i++)` whose bound is absolute, so that only the first search
 * starts from a (w-1)-element sum and every later one from zero. */
int orc_detect_low_quality_regions(const orc_read* r, int window, int quality, int* first, int* last, int cap) {
    int n = 0;
    if (r == NULL || r->len == 0 || window <= 0) return 0;
    const int l = r->len;
    const char* q = r->qual + r->start;
    int start = 0;
    while (start + window <= l) {
        int total = 0;
        for (int i = start; i < window - 1 && i < l; i++) total += q[i];
        int ws = -1;
        for (int s = start; s + window < l; s++) {
            if (total < (33 + quality) * window) {
                ws = s;
                break;
            }
            total += q[s + window];
            total -= q[s];
        }
        if (ws == -1) break;
        int e;
        for (e = ws; e + window < l; e++) {
            total += q[e + window];
            total -= q[e];
            if (total >= (33 + quality) * window) break;
        }
        if (n < cap) {
            first[n] = ws;
            last[n] = e + window - 1;
        }
        n++;
        start = e + window;
    }
    return n;
}

void orc_fraglist_free(orc_fraglist* l) {
    free(l->frag);
    free(l->reg);
    memset(l, 0, sizeof(*l));
}
static fpl_fragment* fraglist_add(orc_fraglist* l) {
    if (l->n_frag == l->cap_frag) {
        l->cap_frag = l->cap_frag ? 2 * l->cap_frag : 1024;
        l->frag = (fpl_fragment*)realloc(l->frag, l->cap_frag * sizeof(fpl_fragment));
    }
    fpl_fragment* f = &l->frag[l->n_frag++];
    memset(f, 0, sizeof(*f));
    return f;
}
static void fraglist_add_region(orc_fraglist* l, uint32_t start, uint32_t len) {
    if (l->n_reg == l->cap_reg) {
        l->cap_reg = l->cap_reg ? 2 * l->cap_reg : 1024;
        l->reg = (fpl_region*)realloc(l->reg, l->cap_reg * sizeof(fpl_region));
    }
    l->reg[l->n_reg].start = start;
    l->reg[l->n_reg].len = len;
    l->n_reg++;
}

/* One output read of the --break / --mask stage (src/seprocessor.cpp:234-281): a window on the original read
 * plus how it got its name. */
typedef struct out_read {
    orc_read r;
    int kind;     /* 0 / 1 / 2, see fpl_fragment */
    int break_no; /* i of the "r<i>-" prefix, 0 = none */
} out_read;

/* passFilter + counters + statRead for the output reads of one input read; masking (Read::maskRegionWithN,
 * src/read.cpp:217-225) happens on a private copy of the fragment's bases, as the reference's strings are. */
static void finish_break_mask(const orc_config* cfg, out_read* outs, int n_out, int64_t* counters, uint32_t C,
                              uint32_t read_index, orc_fraglist* list) {
    const fpl_options* o = &cfg->opt;
    int64_t* post = counters + FPL_OFF_POST(C);
    int64_t* fr = counters + FPL_OFF_FR(C);
    for (int i = 0; i < n_out; i++) {
        orc_read fr_read = outs[i].r;
        char* masked = NULL;
        fpl_fragment* f = fraglist_add(list);
        f->read = read_index;
        f->seq_no = (uint32_t)i;
        f->start = (uint32_t)outs[i].r.start;
        f->len = (uint32_t)outs[i].r.len;
        f->kind = (uint8_t)outs[i].kind;
        f->break_no = (uint16_t)outs[i].break_no;
        f->region_first = list->n_reg;
        if (o->mask_enabled && outs[i].r.len > 0) { /* :254-262 */
            int cap = outs[i].r.len / 2 + 2;
            int* a = (int*)malloc(sizeof(int) * 2 * (size_t)cap);
            int* b = a + cap;
            int nr = orc_detect_low_quality_regions(&outs[i].r, o->mask_window, o->mask_quality, a, b, cap);
            if (nr > 0) {
                masked = (char*)malloc((size_t)outs[i].r.len);
                memcpy(masked, outs[i].r.seq + outs[i].r.start, (size_t)outs[i].r.len);
                for (int j = 0; j < nr; j++) { /* maskRegionWithN(first, last - first + 1) */
                    int st = a[j], ln = b[j] - a[j] + 1;
                    if (st < 0 || ln <= 0 || st >= outs[i].r.len) continue;
                    if (st + ln > outs[i].r.len) ln = outs[i].r.len - st;
                    memset(masked + st, 'N', (size_t)ln);
                    fraglist_add_region(list, (uint32_t)(outs[i].r.start + st), (uint32_t)ln);
                }
                fr_read.seq = masked - outs[i].r.start; /* same window coordinates, private bases */
            }
            free(a);
        }
        f->region_count = list->n_reg - f->region_first;
        int result = orc_pass_filter(&fr_read, o);
        fr[FPL_FR_FILTER + result] += 1;
        f->code = (uint8_t)result;
        if (result == FPL_PASS_FILTER) orc_stat_read(post, C, &fr_read, &f->median_q);
        free(masked);
    }
}

void orc_process_read_ex(const orc_config* cfg, const char* seq, const char* qual, int len,
                         int64_t* counters, uint32_t C, fpl_read_result* res, uint32_t read_index, orc_fraglist* list) {
    const fpl_options* o = &cfg->opt;
    int nad = 2 + cfg->n_fasta;
    int64_t* pre = counters + FPL_OFF_PRE(C);
    int64_t* post = counters + FPL_OFF_POST(C);
    int64_t* fr = counters + FPL_OFF_FR(C);
    int64_t* keyh = counters + FPL_OFF_KEYHIST(C);
    (void)nad;
    memset(res, 0, sizeof(*res));

    orc_read or1 = {seq, qual, 0, len};
    orc_stat_read(pre, C, &or1, &res->median_q_pre); /* :192 */

    orc_read r1 = or1;
    int frontTrimmed = 0;
    int alive = (orc_trim_and_cut(&r1, o, &frontTrimmed) == 0); /* :196 */
    if (!alive) {
        res->dropped = 1;
        return; /* outReads stays empty: no filter result, no output (:228-232,265) */
    }
    if (o->polyx) { /* :198-201 */
        int poly, tl;
        if (orc_trim_polyx(&r1, o->polyx_min_len, &poly, &tl)) {
            fr[FPL_FR_POLYX_READS + poly] += 1;
            fr[FPL_FR_POLYX_BASES + poly] += tl;
        }
    }
    orc_read frags[2];
    int kinds[2] = {0, 0};
    int nfrag = 0;
    if (o->adapter_enabled) { /* :205-229 */
        int trimmed = 0, kl;
        if (cfg->start_len > 0) {
            trimmed += orc_trim_start(&r1, cfg->start_adapter, cfg->start_len, o->ed_max, o->trimming_extension, &kl);
            if (kl > 0) keyh[(0 * 2 + 0) * FPL_KEY_STRIDE + kl]++;
        }
        if (cfg->end_len > 0) {
            trimmed += orc_trim_end(&r1, cfg->end_adapter, cfg->end_len, o->ed_max, o->trimming_extension, &kl);
            if (kl > 0) keyh[(1 * 2 + 1) * FPL_KEY_STRIDE + kl]++;
        }
        if (cfg->n_fasta > 0) { /* trimByMultiSequences, src/adaptertrimmer.cpp:42-57 */
            for (int i = 0; i < cfg->n_fasta; i++) {
                trimmed += orc_trim_start(&r1, cfg->fasta[i].seq, cfg->fasta[i].len, o->ed_max, o->trimming_extension, &kl);
                if (kl > 0) keyh[((2 + i) * 2 + 0) * FPL_KEY_STRIDE + kl]++;
                trimmed += orc_trim_end(&r1, cfg->fasta[i].seq, cfg->fasta[i].len, o->ed_max, o->trimming_extension, &kl);
                if (kl > 0) keyh[((2 + i) * 2 + 1) * FPL_KEY_STRIDE + kl]++;
            }
        }
        if (trimmed > 0) { /* :214-216 addReadTrimmed */
            fr[FPL_FR_ADAPTER_BASES] += trimmed;
            fr[FPL_FR_ADAPTER_READS] += 1;
        }
        int start = -1, glen = 0;
        if (orc_find_middle(&r1, cfg->start_adapter, cfg->start_len, cfg->end_adapter, cfg->end_len,
                            o->ed_max, o->trimming_extension, &start, &glen)) {
            /* Read::breakByGap, src/read.cpp:192-215 */
            int len1 = start;
            int len2 = r1.len - start - glen;
            if (len1 > 0) {
                frags[nfrag].seq = seq; frags[nfrag].qual = qual;
                frags[nfrag].start = r1.start; frags[nfrag].len = len1;
                kinds[nfrag++] = 1;
            }
            if (len2 > 0) {
                frags[nfrag].seq = seq; frags[nfrag].qual = qual;
                frags[nfrag].start = r1.start + start + glen; frags[nfrag].len = len2;
                kinds[nfrag++] = 2;
            }
        } else {
            frags[nfrag] = r1;
            kinds[nfrag++] = 0;
        }
    } else {
        frags[nfrag] = r1;
        kinds[nfrag++] = 0;
    }
    res->r1_start = (uint32_t)r1.start;
    res->r1_len = (uint32_t)r1.len;
    if (o->break_enabled || o->mask_enabled) { /* :234-262, then :265-281 over however many reads result */
        int n_out = 0, cap_out = 8;
        out_read* outs = (out_read*)malloc(sizeof(out_read) * (size_t)cap_out);
        for (int i = 0; i < nfrag; i++) {
            int nr = 0, *ra = NULL, *rb = NULL;
            if (o->break_enabled) {
                int cap = frags[i].len / 2 + 2;
                ra = (int*)malloc(sizeof(int) * 2 * (size_t)cap);
                rb = ra + cap;
                nr = orc_detect_low_quality_regions(&frags[i], o->break_window, o->break_quality, ra, rb, cap);
            }
            if (n_out + nr + 2 > cap_out) {
                cap_out = 2 * (n_out + nr + 2);
                outs = (out_read*)realloc(outs, sizeof(out_read) * (size_t)cap_out);
            }
            if (nr > 0) { /* Read::breakByRegions, src/read.cpp:227-262 */
                const int L = frags[i].len;
                int lastEnd = -1;
                for (int j = 0; j < nr; j++) {
                    int st = ra[j], en = rb[j];
                    if (st < 0) st = 0;
                    if (en >= L) en = L - 1;
                    if (st > en || st >= L) continue;
                    if (st > lastEnd + 1) {
                        out_read* x = &outs[n_out++];
                        x->r = frags[i];
                        x->r.start = frags[i].start + lastEnd + 1;
                        x->r.len = st - lastEnd - 1;
                        x->kind = kinds[i];
                        x->break_no = j + 1;
                    }
                    lastEnd = en;
                }
                if (lastEnd < L - 1) {
                    out_read* x = &outs[n_out++];
                    x->r = frags[i];
                    x->r.start = frags[i].start + lastEnd + 1;
                    x->r.len = L - lastEnd - 1;
                    x->kind = kinds[i];
                    x->break_no = nr + 1;
                }
            } else {
                out_read* x = &outs[n_out++];
                x->r = frags[i];
                x->kind = kinds[i];
                x->break_no = 0;
            }
            free(ra);
        }
        res->n_frag = (uint8_t)(n_out > 255 ? 255 : n_out);
        finish_break_mask(cfg, outs, n_out, counters, C, read_index, list);
        free(outs);
        return;
    }
    res->n_frag = (uint8_t)nfrag;
    for (int i = 0; i < nfrag; i++) { /* :265-281 */
        int result = orc_pass_filter(&frags[i], o);
        fr[FPL_FR_FILTER + result] += 1;
        res->frag_start[i] = (uint32_t)frags[i].start;
        res->frag_len[i] = (uint32_t)frags[i].len;
        res->code[i] = (uint8_t)result;
        res->kind[i] = (uint8_t)kinds[i];
        if (result == FPL_PASS_FILTER) orc_stat_read(post, C, &frags[i], &res->median_q_post[i]);
    }
}

void orc_process_read(const orc_config* cfg, const char* seq, const char* qual, int len,
                      int64_t* counters, uint32_t C, fpl_read_result* res) {
    orc_process_read_ex(cfg, seq, qual, len, counters, C, res, 0, NULL);
}

void orc_process_batch_ex(const orc_config* cfg, const uint8_t* seq, const uint8_t* qual,
                          const uint64_t* off, uint32_t n_reads, int64_t* counters, uint32_t C,
                          fpl_read_result* res, orc_fraglist* list) {
    for (uint32_t i = 0; i < n_reads; i++)
        orc_process_read_ex(cfg, (const char*)seq + off[i], (const char*)qual + off[i],
                            (int)(off[i + 1] - off[i]), counters, C, &res[i], i, list);
}

void orc_process_batch(const orc_config* cfg, const uint8_t* seq, const uint8_t* qual,
                       const uint64_t* off, uint32_t n_reads, int64_t* counters, uint32_t C,
                       fpl_read_result* res) {
    orc_process_batch_ex(cfg, seq, qual, off, n_reads, counters, C, res, NULL);
}

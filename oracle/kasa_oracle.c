/*
 * kasa_oracle.c -- CPU restatement of kASA's `identify` hot path (encode -> sort -> prefix range ->
 * sorted-index merge -> score).  TEST INFRASTRUCTURE ONLY (see kasa_oracle.h): the checker for the
 * HIP path and the "port" CPU baseline of bench.py.  Never linked into, or called from, the product.
 *
 * Written from the behaviour of the reference (kASA v1.4.9); no reference source is copied.  Each
 * function names the reference lines it follows.  Pinned against outputs of the reference's shipped
 * binary by tests/test_oracle_golden.py (fixtures: tests/golden/, generator: make_fixtures.py).
 *
 * Build: see oracle/Makefile (plain gcc, -O2 -ffp-contract=off: IEEE float/double, no fused ops).
 */
#include "kasa_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * A1. Codon -> letter.  kASA.hpp:69-87 (index arithmetic), kASA.hpp:621-667 (table contents).
 * Base code b = (c & 14) >> 1: A,C,T,G,X,Z -> 0..5 (case-insensitive).  LUT index = b0*64+b1*8+b2.
 * Standard genetic code, stops TAA/TAG -> '[' and TGA -> ']', any Z -> '_', else any X -> '^'.
 * The letter code stored is (char & 31).
 * ---------------------------------------------------------------------------------------------- */
void ko_codon_table(uint8_t lut[366])
{
    /* standard code in TCAG order (NCBI transl_table 1) */
    static const char std_tcag[65] =
        "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const int tcag_of_code[4] = { 2 /*A*/, 1 /*C*/, 0 /*T*/, 3 /*G*/ };
    memset(lut, 0, 366); /* ' ' & 31 == 0 in the gaps */
    for (int b0 = 0; b0 < 6; ++b0)
        for (int b1 = 0; b1 < 6; ++b1)
            for (int b2 = 0; b2 < 6; ++b2) {
                char aa;
                if (b0 == 5 || b1 == 5 || b2 == 5) aa = '_';
                else if (b0 == 4 || b1 == 4 || b2 == 4) aa = '^';
                else {
                    int idx = tcag_of_code[b0] * 16 + tcag_of_code[b1] * 4 + tcag_of_code[b2];
                    aa = std_tcag[idx];
                    if (aa == '*') aa = (idx == 14) ? ']' : '[';
                }
                lut[b0 * 64 + b1 * 8 + b2] = (uint8_t)(aa & 31);
            }
}

int ko_codon_table_from_file(const char *path, const char *id, uint8_t lut[366])
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char needle[128], line[6][512];
    snprintf(needle, sizeof(needle), "  id %s ,", id);
    int found = 0;
    while (fgets(line[0], sizeof(line[0]), f))
        if (strstr(line[0], needle)) { found = 1; break; }
    if (found) {
        for (int i = 1; i <= 5; ++i)
            if (!fgets(line[i], sizeof(line[i]), f)) { fclose(f); return 0; }
        /* line 1: amino acids after the first '"'; line 2: start codons (ignored); lines 3-5: Base1..Base3 */
        const char *aa = strchr(line[1], '"');
        const char *b1 = line[3] + strcspn(line[3], "TGCA");
        if (!aa || !*b1) { fclose(f); return 0; }
        ++aa;
        const size_t o = (size_t)(b1 - line[3]);
        for (size_t i = 0; line[3][o + i] && line[3][o + i] != '\n' && line[3][o + i] != '\r'; ++i) {
            const int idx = ((line[3][o + i] & 14) << 5) | ((line[4][o + i] & 14) << 2) | ((line[5][o + i] & 14) >> 1);
            const char c = aa[i];
            if (idx < 366) lut[idx] = (uint8_t)(((c == '*') ? '[' : c) & 31);
        }
    }
    fclose(f);
    return found;
}

float ko_weight(int k) /* Compare.hpp:392: k*k / 625.f, both operands float */
{
    return (float)(k * k) / 625.f;
}

/* ------------------------------------------------------------------------------------------------
 * A3. Read preparation.  Read.hpp:1068-1078 (marker), 633-654 (padding), 36-57 (count).
 * ---------------------------------------------------------------------------------------------- */
static int64_t marker_len(const ko_params *p) /* Read.hpp:1068-1078: '^' x (K-kLow) for protein, 'X' x 3(K-kLow) else */
{
    return (p->protein ? 1 : 3) * (int64_t)(p->K - p->kLow);
}

int64_t ko_padded_len(int64_t rawLen, const ko_params *p)
{
    const int64_t m = marker_len(p);
    int64_t len = rawLen;
    if (len > 0) {
        if (p->protein) {
            while (len + m < p->K) ++len;
        } else if (p->frames == 1) {
            while ((len + m) / 3 < p->K) ++len;
        } else {
            while (len + m < 3 * (int64_t)p->K) ++len;
        }
    }
    return len + m;
}

int64_t ko_kmer_count(int64_t L, const ko_params *p)
{
    if (p->protein) return (L > p->K + 1) ? L - p->K + 1 : 0;
    if (p->frames == 1) {
        const int64_t t = L / 3;
        return (t > p->K + 1) ? t - p->K + 1 : 0;
    }
    return (L > 3 * (int64_t)p->K + 1) ? L - 3 * (int64_t)p->K + 1 : 0;
}

static inline int is_acgt(uint8_t c)
{
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
}

static inline uint8_t letter_at(const uint8_t *s, int64_t pos, const uint8_t *lut)
{
    const int idx = ((s[pos] & 14) << 5) | ((s[pos + 1] & 14) << 2) | ((s[pos + 2] & 14) >> 1);
    return lut[idx];
}

/* Read.hpp:84-220: one k-mer per window start i = 0..count-1, window = K codons from i (3 frames
 * interleaved by the rolling update; emission order is the start position). */
static void encode_prepared(const uint8_t *s, int64_t L, const ko_params *p, const uint8_t *lut,
                            ko_key *out, int64_t count)
{
    const int K = p->K;
    if (p->protein) { /* Read.hpp:60-81 + kASA.hpp:333-379: the letters are the input, code = char & 31 */
        for (int64_t i = 0; i < count; ++i) {
            ko_key v = 0;
            for (int j = 0; j < K; ++j) v = (v << 5) | (ko_key)(s[i + j] & 31);
            out[i] = v;
        }
        return;
    }
    if (p->frames == 1) { /* Read.hpp:223-261: translate frame 0 once, slide over letters */
        for (int64_t i = 0; i < count; ++i) {
            ko_key v = 0;
            for (int j = 0; j < K; ++j) v = (v << 5) | letter_at(s, 3 * (i + j), lut);
            out[i] = v;
        }
        return;
    }
    (void)L;
    for (int64_t i = 0; i < count; ++i) {
        ko_key v = 0;
        for (int j = 0; j < K; ++j) v = (v << 5) | letter_at(s, i + 3 * j, lut);
        out[i] = v;
    }
}

int64_t ko_encode_batch(const uint8_t *bases, const int64_t *off, int64_t nReads, const ko_params *p,
                        const uint8_t lut[366], ko_key *outKmer, uint32_t *outRead)
{
    int64_t total = 0;
    int64_t cap = 0;
    uint8_t *buf = NULL, *rc = NULL;
    const int64_t m = marker_len(p);
    for (int64_t r = 0; r < nReads; ++r) {
        const int64_t raw = off[r + 1] - off[r];
        const int64_t L = ko_padded_len(raw, p);
        const int64_t cnt = (raw > 0) ? ko_kmer_count(L, p) : 0;
        const int strands = (p->frames == 6 && !p->protein) ? 2 : 1; /* kASA.hpp:181: protein switches --six off */
        if (!outKmer) { total += cnt * strands; continue; }
        if (raw == 0) continue;
        if (L + 4 > cap) {
            cap = 2 * (L + 4);
            buf = (uint8_t *)realloc(buf, (size_t)cap);
            rc = (uint8_t *)realloc(rc, (size_t)cap);
        }
        /* Read.hpp:657-675: everything but ACGTacgt becomes Z; Read.hpp:648-650: pad with X */
        const uint8_t *src = bases + off[r];
        if (p->protein) { /* Read.hpp:663-667: '*' becomes '['; padding and marker are '^' (Read.hpp:636-640,1069-1073) */
            for (int64_t i = 0; i < raw; ++i) buf[i] = (src[i] == '*') ? (uint8_t)'[' : src[i];
            for (int64_t i = raw; i < L; ++i) buf[i] = '^';
        } else {
            for (int64_t i = 0; i < raw; ++i) buf[i] = is_acgt(src[i]) ? src[i] : (uint8_t)'Z';
            for (int64_t i = raw; i < L; ++i) buf[i] = 'X'; /* padding + marker are both X */
        }
        if (cnt > 0) {
            encode_prepared(buf, L, p, lut, outKmer + total, cnt);
            for (int64_t i = 0; i < cnt; ++i) outRead[total + i] = (uint32_t)r;
            total += cnt;
        }
        if (strands == 2) {
            /* kASA.hpp:214-221 on the padded read, marker appended afterwards (Read.hpp:612-622) */
            static const uint8_t comp[8] = { 'T', 'G', 'A', 'C', 'X', 'Z', 'X', 'X' };
            const int64_t body = L - m;
            for (int64_t j = 0; j < body; ++j) rc[body - 1 - j] = comp[(buf[j] >> 1) & 7];
            for (int64_t i = body; i < L; ++i) rc[i] = 'X';
            if (cnt > 0) {
                encode_prepared(rc, L, p, lut, outKmer + total, cnt);
                for (int64_t i = 0; i < cnt; ++i) outRead[total + i] = (uint32_t)r;
                total += cnt;
            }
        }
    }
    free(buf);
    free(rc);
    return total;
}

/* ------------------------------------------------------------------------------------------------
 * A5. Sort by k-mer (Compare.hpp:1077; unstable there, stable here -- tie order never reaches the
 * output, DESIGN.md "Oracle") and prefix ranges (Compare.hpp:1098-1117, Trie.hpp:398-462,494-520).
 * ---------------------------------------------------------------------------------------------- */
void ko_sort_queries(ko_key *kmer, uint32_t *read, uint64_t n)
{
    if (n < 2) return;
    ko_key *k2 = (ko_key *)malloc(n * sizeof(ko_key));
    uint32_t *r2 = (uint32_t *)malloc(n * sizeof(uint32_t));
    ko_key *ka = kmer, *kb = k2;
    uint32_t *ra = read, *rb = r2;
    for (int pass = 0; pass < (int)sizeof(ko_key); ++pass) {
        const int sh = 8 * pass;
        uint64_t hist[257];
        memset(hist, 0, sizeof(hist));
        for (uint64_t i = 0; i < n; ++i) ++hist[(uint32_t)((ka[i] >> sh) & 255) + 1];
        int trivial = 0;
        for (int d = 0; d < 256; ++d)
            if (hist[d + 1] == n) trivial = 1;
        if (trivial) continue;
        for (int d = 0; d < 256; ++d) hist[d + 1] += hist[d];
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t dst = hist[(uint32_t)((ka[i] >> sh) & 255)]++;
            kb[dst] = ka[i];
            rb[dst] = ra[i];
        }
        ko_key *tk = ka; ka = kb; kb = tk;
        uint32_t *tr = ra; ra = rb; rb = tr;
    }
    if (ka != kmer) {
        memcpy(kmer, ka, n * sizeof(ko_key));
        memcpy(read, ra, n * sizeof(uint32_t));
    }
    free(k2);
    free(r2);
}

void ko_ranges(const ko_index *ix, const ko_params *p, const ko_key *kmer, uint64_t n,
               uint64_t *rangeStart, uint32_t *rangeLenM1)
{
    const int sh = 5 * (p->K - 6);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t pre = (uint64_t)(kmer[i] >> sh);   /* Compare.hpp:1102-1107: >> 30, or >> 95 for 128-bit keys */
        uint64_t lo = 0, hi = ix->nTrie;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ix->triePrefix[mid] < pre) lo = mid + 1; else hi = mid;
        }
        if (lo < ix->nTrie && ix->triePrefix[lo] == pre) {
            rangeStart[i] = ix->trieStart[lo];
            rangeLenM1[i] = ix->trieLenM1[lo];
        } else {
            rangeStart[i] = KO_RANGE_NONE;
            rangeLenM1[i] = 0;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Per-level state shared by both comparison routines: the remembered prefix, the read list of the
 * open group and its taxon set (BitArray.hpp:98-117: bitset + insertion-ordered list).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    ko_key mem;       /* vMemoryOfSeenkMers */
    uint64_t hits;    /* vPositions */
    uint64_t *reads;  /* vReadIDs */
    uint64_t readsCap;
    uint64_t *bits;   /* taxon bitset */
    uint32_t *taxList;
    uint32_t nTax;
} level_state;

typedef struct {
    const ko_params *p;
    const ko_index *ix;
    int nK;
    level_state *lv;
    double *countAll;
    uint64_t *countUnique;
    uint64_t *countTotal;
    float *M;
    int haveM;
} cmp_ctx;

static void ctx_init(cmp_ctx *c, const ko_params *p, const ko_index *ix, double *ca, uint64_t *cu,
                     uint64_t *ct, float *M)
{
    c->p = p; c->ix = ix; c->nK = p->kHigh - p->kLow + 1;
    c->countAll = ca; c->countUnique = cu; c->countTotal = ct; c->M = M; c->haveM = (M != NULL);
    c->lv = (level_state *)calloc((size_t)c->nK, sizeof(level_state));
    const size_t words = ((size_t)ix->nTaxa + 63) >> 6;
    for (int i = 0; i < c->nK; ++i) {
        c->lv[i].bits = (uint64_t *)calloc(words ? words : 1, 8);
        c->lv[i].taxList = (uint32_t *)malloc(((size_t)ix->nTaxa + 1) * 4);
        c->lv[i].readsCap = 128;
        c->lv[i].reads = (uint64_t *)malloc(128 * 8);
    }
}

static void ctx_free(cmp_ctx *c)
{
    for (int i = 0; i < c->nK; ++i) { free(c->lv[i].bits); free(c->lv[i].taxList); free(c->lv[i].reads); }
    free(c->lv);
}

static inline void lv_mark(level_state *s, uint32_t t)
{
    if (!((s->bits[t >> 6] >> (t & 63)) & 1)) {
        s->taxList[s->nTax++] = t;
        s->bits[t >> 6] |= 1ULL << (t & 63);
    }
}

static inline void lv_clear_taxa(level_state *s)
{
    for (uint32_t i = 0; i < s->nTax; ++i) s->bits[s->taxList[i] >> 6] = 0;
    s->nTax = 0;
}

static inline void lv_push(cmp_ctx *c, level_state *s, uint32_t rid)
{
    if (c->haveM) { /* Compare.hpp:721-728; profile-only mode just counts (687-689) */
        if (s->hits >= s->readsCap) {
            s->readsCap *= 2;
            s->reads = (uint64_t *)realloc(s->reads, s->readsCap * 8);
        }
        s->reads[s->hits] = rid;
    }
    s->hits++;
}

/* A7. Compare.hpp:648-673 -> 516-532 (non-AVX), 534-597 (AVX, only when avxQuirk && n > 3),
 * 599-645 (counts only).  lvIdx = 0 for kHigh (Compare.hpp:922-925). */
static void flush_level(cmp_ctx *c, int lvIdx)
{
    level_state *s = &c->lv[lvIdx];
    const uint64_t n = s->nTax;
    const uint64_t h = s->hits;
    if (n == 0) return; /* the reference iterates an empty taxon list: nothing is touched */
    const uint32_t nTaxa = c->ix->nTaxa;
    const uint64_t base = (uint64_t)nTaxa * (uint64_t)lvIdx;
    const float weight = ko_weight(c->p->kHigh - lvIdx);
    const float score = weight * (1.f / (float)n);
    const double counts = (double)h / (double)n;
    const int avx = c->p->avxQuirk && n > 3;

    if (!c->haveM || !avx) {
        for (uint32_t i = 0; i < s->nTax; ++i) {
            const uint32_t t = s->taxList[i];
            c->countAll[base + t] += counts;
            if (c->p->coverage && c->countTotal) c->countTotal[base + t] += 1;
            if (n == 1 && !avx) c->countUnique[base + t] += h;
            if (c->haveM)
                for (uint64_t r = 0; r < h; ++r) {
                    float *cell = &c->M[s->reads[r] * (uint64_t)nTaxa + t];
                    *cell = *cell + score;
                }
        }
        return;
    }
    /* scoreMatchAVX: 8-slot gather blocks; a taxon takes reads from the list head only while the
     * block has room, a block is written back slot by slot (Compare.hpp:559-575,588-596). */
    uint64_t blkRead[8], blkTax[8];
    float blkVal[8];
    uint32_t blk = 0;
    for (uint32_t i = 0; i < s->nTax; ++i) {
        const uint32_t t = s->taxList[i];
        c->countAll[base + t] += counts;
        if (c->p->coverage && c->countTotal) c->countTotal[base + t] += 1;
        for (uint64_t r = 0; blk < 8 && r < h; ++r, ++blk) {
            blkRead[blk] = s->reads[r];
            blkTax[blk] = t;
            blkVal[blk] = c->M[s->reads[r] * (uint64_t)nTaxa + t];
        }
        if (blk == 8) {
            for (uint32_t j = 0; j < 8; ++j) blkVal[j] = blkVal[j] + score;
            for (uint32_t j = 0; j < 8; ++j) c->M[blkRead[j] * (uint64_t)nTaxa + blkTax[j]] = blkVal[j];
            blk = 0;
        }
    }
    if (blk > 0) {
        for (uint32_t j = 0; j < blk; ++j) blkVal[j] = blkVal[j] + score;
        for (uint32_t j = 0; j < blk; ++j) c->M[blkRead[j] * (uint64_t)nTaxa + blkTax[j]] = blkVal[j];
    }
}

static inline int shift_of(const ko_params *p, int lvIdx) { return 5 * (p->K - (p->kHigh - lvIdx)); }

/* first position in [lo, hiIncl+1) whose (kmer >> sh) is >= val (std::lower_bound, Compare.hpp:824,980) */
static uint64_t lower_bound_shifted(const ko_key *km, uint64_t lo, uint64_t hiExcl, int sh, ko_key val)
{
    while (lo < hiExcl) {
        const uint64_t mid = lo + ((hiExcl - lo) >> 1);
        if ((km[mid] >> sh) < val) lo = mid + 1; else hiExcl = mid;
    }
    return lo;
}

/* ------------------------------------------------------------------------------------------------
 * A6. Compare.hpp:678-1069, statement by statement in the same order (64-bit keys, no spaced masks,
 * no post-processing).  Level index lv: 0 = kHigh ... nK-1 = kLow (kASA.hpp:299-302).
 * ---------------------------------------------------------------------------------------------- */
static int compare_sequential_impl(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                          const uint32_t *qRead, const uint64_t *qRS, const uint32_t *qRL,
                          uint64_t nQ, uint64_t nReads, double *countAll, uint64_t *countUnique,
                          uint64_t *countTotal, float *M, uint8_t *ml)
{
    /* ml: --coherence (Compare.hpp:847-848,882-884,912-914,948): wherever the read id of a query joins the hit list of a
     * level, setMatchLength overwrites the query's match length with that level's k; the last write stands. */
#define KO_ML(level) do { if (ml) ml[qi] = (uint8_t)(p->kHigh - (level)); } while (0)
    (void)nReads;
    cmp_ctx c;
    ctx_init(&c, p, ix, countAll, countUnique, countTotal, M);
    const int nK = c.nK;
    const int low = nK - 1;
    const ko_key *km = ix->kmer;
    /* the reference's `compare` functor (Compare.hpp:700-706) is declared with uint64_t parameters: with 128-bit keys
     * it sees the low words only.  EQ/CMP3 are that functor; everything else in the routine is full width. */
    const int q64 = (sizeof(ko_key) > 8) && p->cmp64Quirk;
#define KO_EQ(a, b) (q64 ? ((uint64_t)(a) == (uint64_t)(b)) : ((a) == (b)))
#define KO_LT(a, b) (q64 ? ((uint64_t)(a) < (uint64_t)(b)) : ((a) < (b)))
    const uint32_t *tx = ix->tax;

    uint64_t pos = 0;
    while (pos < nQ) {
        const uint64_t rs = qRS[pos];
        const uint64_t rl = qRL[pos];
        uint64_t qi = pos;
        if (rs == KO_RANGE_NONE) { ++pos; continue; } /* :753-756 */
        while (pos < nQ) {                             /* :758-766 */
            const uint64_t cur = qRS[pos];
            if (cur != rs && cur != KO_RANGE_NONE) break;
            ++pos;
        }
        for (int i = 0; i < nK; ++i) {                 /* :769-773 */
            c.lv[i].hits = 0; c.lv[i].mem = 0; lv_clear_taxa(&c.lv[i]);
        }
        ko_key seenKmer = 0;                           /* :774 */
        const uint64_t rb = rs, re = rs + rl;          /* :777-778, re is the last entry (inclusive) */
        uint64_t it = rb;
        int determine = 1;

        for (; qi < pos; ++qi) {                       /* :785 */
            if (qRS[qi] == KO_RANGE_NONE) continue;
            int sh = shift_of(p, low);
            const ko_key q = qKmer[qi];
            const uint32_t rid = qRead[qi];
            ko_key qs = q >> sh;
            int inputIterated = 1;

            if (seenKmer != q && (km[it] >> sh) != qs && determine) { /* :803-829 */
                if ((km[rb] >> sh) == qs) {
                    it = rb;
                } else if ((km[re] >> sh) == qs) {
                    uint64_t t = 1;
                    while ((km[re - t] >> sh) == qs) ++t;
                    it = re - (t - 1);
                } else if (qs < (km[rb] >> sh) || qs > (km[re] >> sh)) {
                    continue; /* not inside the range; `determine` stays set (:819) */
                } else {
                    it = lower_bound_shifted(km, rb, re + 1, sh, qs);
                }
            }
            determine = 0;

            if ((qs & 31) == 30) continue;             /* :836-838 */

            if (KO_EQ(seenKmer, q) || it == re + 1) {  /* :841-853 duplicates / index exhausted */
                for (int l = low; l >= 0; --l)
                    if (KO_EQ(q >> shift_of(p, l), c.lv[l].mem)) { lv_push(&c, &c.lv[l], rid); KO_ML(l); }
                continue;
            }
            seenKmer = q;                              /* :855 */

            int breakOut = 0;
            while (it != re + 1 && !breakOut) {        /* :861 */
                const ko_key e = km[it];
                int l = low;
                for (; l >= 0; --l) {                  /* :865 */
                    sh = shift_of(p, l);
                    qs = q >> sh;
                    const ko_key es = e >> sh;
                    if (KO_LT(qs, es)) {               /* :874-893 input smaller */
                        if (inputIterated)
                            for (int u = l; u >= 0; --u) {
                                if (KO_EQ(q >> shift_of(p, u), c.lv[u].mem)) { lv_push(&c, &c.lv[u], rid); KO_ML(u); }
                                else break;
                            }
                        breakOut = 1;
                        break;
                    } else if (!KO_LT(es, qs)) {       /* :895-956 equal (compareTwoKmers: neither is smaller) */
                        if ((qs & 31) == 30) { breakOut = 1; break; }
                        level_state *s = &c.lv[l];
                        if (KO_EQ(qs, s->mem)) {
                            lv_mark(s, tx[it]);
                            if (inputIterated) { lv_push(&c, s, rid); KO_ML(l); }
                        } else {
                            flush_level(&c, l);
                            s->hits = 0;
                            lv_push(&c, s, rid);
                            KO_ML(l);
                            lv_clear_taxa(s);
                            lv_mark(s, tx[it]);
                            s->mem = qs;
                        }
                    } else {                           /* :957-993 index smaller: walk / jump */
                        uint64_t t = 1;
                        while (it + t != re + 1) {
                            const ko_key nx = km[it + t];
                            if (qs > (nx >> sh)) {                  /* :963, full width */
                                int u = low;
                                for (; u >= 0; --u) {
                                    if (KO_EQ(c.lv[u].mem, nx >> shift_of(p, u))) lv_mark(&c.lv[u], tx[it + t]);
                                    else break;
                                }
                                if (u < low) {
                                    ++t;
                                } else {
                                    t = lower_bound_shifted(km, it + t, re + 1, sh, qs) - it;
                                    break;
                                }
                            } else {
                                break;
                            }
                        }
                        it += t;
                        break;
                    }
                }
                if (l == -1) ++it;                     /* :997-999 */
                inputIterated = 0;
            }
        }

        uint64_t t = 0;                                /* :1007-1028 rest of the range */
        while (it + t != re + 1) {
            const ko_key nx = km[it + t];
            int u = low;
            for (; u >= 0; --u) {
                if (KO_EQ(c.lv[u].mem, nx >> shift_of(p, u))) lv_mark(&c.lv[u], tx[it + t]);
                else break;
            }
            if (u < low) ++t; else break;
        }
        for (int l = low; l >= 0; --l) flush_level(&c, l); /* :1032-1041 */
    }
#undef KO_EQ
#undef KO_LT
#undef KO_ML
    ctx_free(&c);
    return 0;
}

int ko_compare_sequential(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                          const uint32_t *qRead, const uint64_t *qRS, const uint32_t *qRL,
                          uint64_t nQ, uint64_t nReads, double *countAll, uint64_t *countUnique,
                          uint64_t *countTotal, float *M)
{
    return compare_sequential_impl(p, ix, qKmer, qRead, qRS, qRL, nQ, nReads, countAll, countUnique, countTotal, M, NULL);
}

int ko_compare_sequential_ml(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                             const uint32_t *qRead, const uint64_t *qRS, const uint32_t *qRL,
                             uint64_t nQ, uint64_t nReads, double *countAll, uint64_t *countUnique,
                             uint64_t *countTotal, float *M, uint8_t *matchLen)
{
    return compare_sequential_impl(p, ix, qKmer, qRead, qRS, qRL, nQ, nReads, countAll, countUnique, countTotal, M, matchLen);
}

/* ------------------------------------------------------------------------------------------------
 * the same result as a group-by.  Per prefix range, per query in sorted
 * order, per k ascending: prefix P; '^' ends the query; absent from the range ends the query; same P
 * as the open group of that k joins it, otherwise the open group is flushed and a new one starts
 * with all distinct taxa of the index entries carrying P (index order).  This is what the device
 * kernels compute; tests require it to equal ko_compare_sequential bit for bit.
 * ---------------------------------------------------------------------------------------------- */
static int compare_closed_form_impl(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                           const uint32_t *qRead, const uint64_t *qRS, const uint32_t *qRL,
                           uint64_t nQ, uint64_t nReads, double *countAll, uint64_t *countUnique,
                           uint64_t *countTotal, float *M, uint8_t *depth)
{
    (void)nReads;
    cmp_ctx c;
    ctx_init(&c, p, ix, countAll, countUnique, countTotal, M);
    const int nK = c.nK;
    const int low = nK - 1;
    const ko_key *km = ix->kmer;
    uint64_t pos = 0;
    while (pos < nQ) {
        const uint64_t rs = qRS[pos];
        if (rs == KO_RANGE_NONE) { ++pos; continue; }
        const uint64_t lo = rs, hi = rs + qRL[pos] + 1; /* [lo, hi) */
        uint64_t qi = pos;
        while (pos < nQ && (qRS[pos] == rs || qRS[pos] == KO_RANGE_NONE)) ++pos;
        for (int i = 0; i < nK; ++i) { c.lv[i].hits = 0; c.lv[i].mem = 0; lv_clear_taxa(&c.lv[i]); }
        for (; qi < pos; ++qi) {
            if (qRS[qi] == KO_RANGE_NONE) continue;
            const ko_key q = qKmer[qi];
            uint64_t a = lo, b = hi;
            for (int l = low; l >= 0; --l) {
                const int sh = shift_of(p, l);
                const ko_key P = q >> sh;
                if ((P & 31) == 30) break;
                a = lower_bound_shifted(km, a, b, sh, P);
                b = lower_bound_shifted(km, a, b, sh, P + 1);
                if (a == b) break;
                level_state *s = &c.lv[l];
                if (depth) depth[qi] = (uint8_t)(p->kHigh - l);     /* the deepest matched level: what the device calls d */
                if (s->nTax && s->mem == P) {
                    lv_push(&c, s, qRead[qi]);
                } else {
                    flush_level(&c, l);
                    s->hits = 0;
                    lv_clear_taxa(s);
                    s->mem = P;
                    lv_push(&c, s, qRead[qi]);
                    for (uint64_t i = a; i < b; ++i) lv_mark(s, ix->tax[i]);
                }
            }
        }
        for (int l = low; l >= 0; --l) flush_level(&c, l);
    }
    ctx_free(&c);
    return 0;
}

int ko_compare_closed_form(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                           const uint32_t *qRead, const uint64_t *qRS, const uint32_t *qRL,
                           uint64_t nQ, uint64_t nReads, double *countAll, uint64_t *countUnique,
                           uint64_t *countTotal, float *M)
{
    return compare_closed_form_impl(p, ix, qKmer, qRead, qRS, qRL, nQ, nReads, countAll, countUnique, countTotal, M, NULL);
}

int ko_compare_closed_form_ml(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                              const uint32_t *qRead, const uint64_t *qRS, const uint32_t *qRL,
                              uint64_t nQ, uint64_t nReads, double *countAll, uint64_t *countUnique,
                              uint64_t *countTotal, float *M, uint8_t *depth)
{
    return compare_closed_form_impl(p, ix, qKmer, qRead, qRS, qRL, nQ, nReads, countAll, countUnique, countTotal, M, depth);
}

/* ------------------------------------------------------------------------------------------------
 * --coherence: Compare::postProcess (Compare.hpp:2607-2728), statement by statement.  The elements are the batch's
 * k-mers sorted by (read id, frame, position) (:2609-2630) -- for single-end input that is the order the reader emitted
 * them in (Read.hpp:128-131,172-175,210-213: position = running window number of the strand, frame = 0 forward / 1 reverse
 * complement); len = the match length setMatchLength left (0 = unmatched).  One walk over the whole batch: `rid` is advanced
 * by the loop, not read off the elements, so an element is credited to whatever read the walk believes it is in -- a read
 * without k-mers takes the first element of its successor, and after a strand switch the search for the next match runs on
 * into the following reads (:2712-2722).  scores[] must come in zeroed (:3319).
 * Returns 0, or 1 where the reference throws std::out_of_range (its vector::at at :2667 after the search of :2712-2722 ran
 * off the end: a batch whose last strand switch finds no further match); *failIdx = the index it asked for.
 * ---------------------------------------------------------------------------------------------- */
static inline void coh_set(float *cell, float v) { if (*cell < v) *cell = v; }           /* :2650-2652, std::max */

int ko_coherence(const uint32_t *read, const uint32_t *pos, const uint8_t *frame, const uint8_t *len, uint64_t n,
                 int sixFrames, float *scores, uint64_t nReads, uint64_t *failIdx)
{
    uint32_t rid = 0, last = 0, cur = 0, cnt = 0;                                        /* :2632-2636 */
    uint64_t idx = 0;
    while (idx < n) {                                                                    /* :2637-2648 */
        const uint8_t ml = len[idx];
        if (ml != 0) { rid = read[idx]; last = pos[idx] + ml; ++idx; break; }
        ++idx;
    }
#define KO_DET(next) do { const uint32_t nx_ = (next); if (nx_ > cur) { cur = nx_; cnt = 1; } else if (nx_ == cur) cnt++; } while (0)   /* :2653-2662 */
#define KO_CLUSTER() coh_set(&scores[rid], (float)cur + 1.0f - 1.0f / (float)cnt)        /* 1.0f / 0 = inf: the score stays */
    for (; rid < nReads && idx < n; ++rid) {                                             /* :2665 */
        for (int fb = 0; fb < 1 + (sixFrames ? 1 : 0);) {                                /* :2667 */
            if (idx >= n) { if (failIdx) *failIdx = idx; return 1; }                     /* vIn.getLength -> vector::at throws */
            const uint8_t ml = len[idx];                                                 /* :2670 */
            if (ml != 0) {
                if (pos[idx] <= last) {                                                  /* :2673 */
                    if (pos[idx] + ml < last) KO_DET((uint32_t)ml);                      /* :2674-2676 */
                    else { const int32_t ov = (int32_t)last - (int32_t)pos[idx]; KO_DET((uint32_t)ov); }   /* :2677-2680 */
                } else {                                                                 /* :2683-2687 */
                    KO_CLUSTER();
                    cur = 0;
                }
                last = pos[idx] + ml;                                                    /* :2688 */
            }
            ++idx;                                                                       /* :2691 */
            if (idx == n) { KO_CLUSTER(); break; }                                       /* :2693-2696 */
            if (read[idx] != rid) {                                                      /* :2699-2706 */
                KO_CLUSTER();
                last = UINT32_MAX; cur = 0; cnt = 0;
                break;
            }
            if (frame[idx] != fb) {                                                      /* :2709-2726 */
                KO_CLUSTER();
                cur = 0; cnt = 0;
                ++fb;
                while (idx < n) {
                    const uint8_t m2 = len[idx];
                    if (m2 != 0) { last = pos[idx] + m2; ++idx; break; }
                    ++idx;
                }
            }
        }
    }
#undef KO_DET
#undef KO_CLUSTER
    return 0;
}

uint64_t ko_unique_queries(ko_key *kmer, uint32_t *read, uint64_t n)
{
    if (n == 0) return 0;
    uint64_t w = 1;
    for (uint64_t i = 1; i < n; ++i)
        if (kmer[i] != kmer[w - 1] || read[i] != read[w - 1]) {
            kmer[w] = kmer[i];
            read[w] = read[i];
            ++w;
        }
    return w;
}

/* ------------------------------------------------------------------------------------------------
 * A8. Per-read numbers.  Compare.hpp:1452-1481, 1510, 1634.
 * ---------------------------------------------------------------------------------------------- */
float ko_best_score(uint64_t readLen, const ko_params *p)
{
    float best = 0.f;
    for (int32_t i = p->kLow; i <= p->kHigh; ++i) {
        const float w = ko_weight(i);
        if (p->protein) {
            best += (float)(readLen - (uint64_t)i + 1) * w;
        } else if (p->frames == 1) {
            best += (float)(readLen / 3 - (uint64_t)i + 1) * w;
        } else if (p->frames == 6) {
            best += (float)(2 * (readLen - (uint64_t)(i * 3) + 1)) * w;
        } else {
            best += (float)(readLen - (uint64_t)(i * 3) + 1) * w;
        }
    }
    return best;
}

double ko_relative_score(float kmerScore, uint64_t freqAtKHigh, uint64_t readLen, const ko_params *p)
{
    /* the reference subtracts in 32-bit unsigned arithmetic (uint32_t length, int K) */
    const uint32_t span = (uint32_t)readLen - (uint32_t)(p->protein ? p->K : p->K * 3) + 1u;
    return (double)kmerScore / (1.0 + log2((double)freqAtKHigh * (double)span));
}

double ko_error_score(float bestScore, float kmerScore)
{
    const float e = (bestScore - kmerScore) / bestScore;
    return (double)e;
}

/* ------------------------------------------------------------------------------------------------
 * The batch with the reference's threading model (-n): for bench.py's cpu_baseline.
 * Compare.hpp:3107-3124 + Read.hpp:763-827: reads split over the worker threads for translation;
 * Compare.hpp:1123-1132: parallel sort of the queries (here: stable bucket pass on the top 16 bits, buckets sorted by the
 * threads); Compare.hpp:1098-1117: ranges per thread; Compare.hpp:3263-3283: the sorted queries cut into one slice per
 * thread on prefix-range boundaries, every thread merges its slice with PRIVATE count tables (:922) into the SHARED
 * score matrix (unsynchronised `+=`, as in the reference: per-read floats may differ in their last digits between
 * runs); Compare.hpp:3445-3454: the count tables summed.
 * ------------------------------------------------------------------------------------------------ */
#include <pthread.h>

typedef struct {
    const ko_params *p; const ko_index *ix; const uint8_t *bases; const int64_t *off; const uint8_t *lut;
    int nThreads, tid;
    int64_t r0, r1;               /* reads of this thread */
    int64_t *kcount;              /* k-mers per thread */
    ko_key *km, *km2; uint32_t *rd, *rd2; uint64_t nQ; uint64_t *koff;   /* koff[t] = first query of thread t */
    uint64_t *hist;               /* [nThreads][KO_NB] */
    uint64_t *bucketStart;        /* [KO_NB + 1] */
    int zeroM;                    /* the score matrix still has to be cleared */
    int timing; double t0;
    volatile int *nextBucket;
    uint64_t *rs; uint32_t *rl;
    uint64_t s0, s1;              /* slice of the sorted queries */
    uint64_t nReads; double *ca; uint64_t *cu, *ct; float *M;
    pthread_barrier_t *bar;
    uint64_t *cuts;
} ko_job;

/* Sort buckets = the top KO_BBITS key bits (three letters and a bit): the first letters of amino-acid-like k-mers are far from
 * uniform, and with the 256 buckets of one byte the largest bucket -- sorted by ONE thread -- held a twentieth of the batch,
 * which capped the speed-up of the whole run near 1.6 on a 256-thread host.  65 536 buckets are dealt out dynamically. */
#define KO_BBITS 16
#define KO_NB (1 << KO_BBITS)
static int key_bucket(ko_key k, int K) { return (int)((k >> (5 * K - KO_BBITS)) & (KO_NB - 1)); }

#include <time.h>
#include <stdio.h>
static double ko_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
/* KO_TIMING=1: thread 0 reports when it passed each phase (seconds since the workers started) */
#define KO_MARK(name) do { if (t == 0 && j->timing) fprintf(stderr, "ko_identify_threaded: %-22s %8.3f s\n", name, ko_now() - j->t0); } while (0)

static void *ko_worker(void *arg)
{
    ko_job *j = (ko_job *)arg;
    const int T = j->nThreads, t = j->tid;
    /* 1. translate my reads */
    const int64_t n = j->r1 - j->r0;
    j->kcount[t] = n > 0 ? ko_encode_batch(j->bases, j->off + j->r0, n, j->p, j->lut, NULL, NULL) : 0;
    pthread_barrier_wait(j->bar);
    if (t == 0) { uint64_t run = 0; for (int i = 0; i < T; ++i) { j->koff[i] = run; run += (uint64_t)j->kcount[i]; } j->koff[T] = run; }
    pthread_barrier_wait(j->bar);
    const uint64_t q0 = j->koff[t], q1 = j->koff[t + 1];
    if (n > 0) {
        ko_encode_batch(j->bases, j->off + j->r0, n, j->p, j->lut, j->km + q0, j->rd + q0);
        for (uint64_t i = q0; i < q1; ++i) j->rd[i] += (uint32_t)j->r0;
    }
    pthread_barrier_wait(j->bar);
    KO_MARK("translated");
    /* the score matrix is touched first by the workers, a share each: its pages are spread over the memory nodes and the
     * page faults (3.4 GB at 600 000 reads x 1400 taxa) are not taken one after the other by the thread that merges first */
    if (j->zeroM) {
        const uint64_t cellsM = j->nReads * (uint64_t)j->ix->nTaxa, za = cellsM * (uint64_t)t / (uint64_t)T, ze = cellsM * (uint64_t)(t + 1) / (uint64_t)T;
        memset(j->M + za, 0, (size_t)(ze - za) * 4);
    }
    pthread_barrier_wait(j->bar);
    KO_MARK("matrix cleared");
    /* 2. parallel sort: stable bucket pass on the top bits, then the buckets */
    uint64_t *h = j->hist + (size_t)t * KO_NB;
    memset(h, 0, KO_NB * sizeof(uint64_t));
    for (uint64_t i = q0; i < q1; ++i) ++h[key_bucket(j->km[i], j->p->K)];
    pthread_barrier_wait(j->bar);
    {   /* bucket sizes: every thread sums a share of the buckets over all threads ... */
        const int b0 = KO_NB * t / T, b1 = KO_NB * (t + 1) / T;
        for (int b = b0; b < b1; ++b) { uint64_t c = 0; for (int i = 0; i < T; ++i) c += j->hist[(size_t)i * KO_NB + b]; j->bucketStart[b + 1] = c; }
    }
    pthread_barrier_wait(j->bar);
    if (t == 0) { j->bucketStart[0] = 0; for (int b = 0; b < KO_NB; ++b) j->bucketStart[b + 1] += j->bucketStart[b]; }   /* ... one running sum ... */
    pthread_barrier_wait(j->bar);
    {   /* ... and the threads' places inside the buckets, again a share of the buckets each */
        const int b0 = KO_NB * t / T, b1 = KO_NB * (t + 1) / T;
        for (int b = b0; b < b1; ++b) { uint64_t run = j->bucketStart[b]; for (int i = 0; i < T; ++i) { const uint64_t c = j->hist[(size_t)i * KO_NB + b]; j->hist[(size_t)i * KO_NB + b] = run; run += c; } }
    }
    pthread_barrier_wait(j->bar);
    for (uint64_t i = q0; i < q1; ++i) { const uint64_t d = h[key_bucket(j->km[i], j->p->K)]++; j->km2[d] = j->km[i]; j->rd2[d] = j->rd[i]; }
    pthread_barrier_wait(j->bar);
    KO_MARK("bucket pass");
    for (;;) {                                                     /* runs of 64 buckets at a time */
        const int b = __sync_fetch_and_add(j->nextBucket, 64);
        if (b >= KO_NB) break;
        for (int bb = b; bb < b + 64 && bb < KO_NB; ++bb) {
            const uint64_t a = j->bucketStart[bb], e = j->bucketStart[bb + 1];
            if (e > a + 1) ko_sort_queries(j->km2 + a, j->rd2 + a, e - a);
        }
    }
    pthread_barrier_wait(j->bar);
    KO_MARK("buckets sorted");
    /* 3. ranges of my share */
    const uint64_t nQ = j->nQ, a = nQ * (uint64_t)t / (uint64_t)T, e = nQ * (uint64_t)(t + 1) / (uint64_t)T;
    if (e > a) ko_ranges(j->ix, j->p, j->km2 + a, e - a, j->rs + a, j->rl + a);
    pthread_barrier_wait(j->bar);
    KO_MARK("ranges");
    /* 4. slices on range boundaries (Compare.hpp:3263-3283), private tables, shared matrix */
    if (t == 0) {
        j->cuts[0] = 0;
        for (int i = 1; i < T; ++i) {
            uint64_t c = nQ * (uint64_t)i / (uint64_t)T;
            if (c < j->cuts[i - 1]) c = j->cuts[i - 1];
            while (c > 0 && c < nQ && j->rs[c] == j->rs[c - 1] && j->rs[c] != KO_RANGE_NONE) ++c;
            j->cuts[i] = c;
        }
        j->cuts[T] = nQ;
    }
    pthread_barrier_wait(j->bar);
    const uint64_t s0 = j->cuts[t], s1 = j->cuts[t + 1];
    if (s1 > s0)
        ko_compare_sequential(j->p, j->ix, j->km2 + s0, j->rd2 + s0, j->rs + s0, j->rl + s0, s1 - s0, j->nReads, j->ca, j->cu, j->ct, j->M);
    KO_MARK("my slice merged");
    pthread_barrier_wait(j->bar);
    KO_MARK("all slices merged");
    return NULL;
}

int ko_identify_threaded(const ko_params *p, const ko_index *ix, const uint8_t *bases, const int64_t *off, int64_t nReads,
                         const uint8_t lut[366], int nThreads, double *countAll, uint64_t *countUnique, float *M, uint64_t *nQueries)
{
    if (nThreads < 1) nThreads = 1;
    const int T = nThreads;
    const int64_t total = ko_encode_batch(bases, off, nReads, p, lut, NULL, NULL);   /* sizes only (the reference knows them from its info pass) */
    const uint64_t nQ = (uint64_t)total;
    const int nK = p->kHigh - p->kLow + 1;
    const size_t cells = (size_t)nK * ix->nTaxa;
    ko_key *km = (ko_key *)malloc((nQ + 1) * sizeof(ko_key)), *km2 = (ko_key *)malloc((nQ + 1) * sizeof(ko_key));
    uint32_t *rd = (uint32_t *)malloc((nQ + 1) * 4), *rd2 = (uint32_t *)malloc((nQ + 1) * 4), *rl = (uint32_t *)malloc((nQ + 1) * 4);
    uint64_t *rs = (uint64_t *)malloc((nQ + 1) * 8);
    int64_t *kcount = (int64_t *)calloc((size_t)T, 8);
    uint64_t *koff = (uint64_t *)calloc((size_t)T + 1, 8), *hist = (uint64_t *)malloc((size_t)T * KO_NB * 8), *cuts = (uint64_t *)calloc((size_t)T + 1, 8);
    uint64_t *bucketStart = (uint64_t *)calloc((size_t)KO_NB + 1, 8);
    double *ca = (double *)calloc(cells * (size_t)T, 8);
    uint64_t *cu = (uint64_t *)calloc(cells * (size_t)T, 8), *ct = (uint64_t *)calloc(cells * (size_t)T, 8);
    float *Mloc = M ? M : (float *)malloc((size_t)nReads * ix->nTaxa * 4 + 4);   /* cleared by the workers (first touch) */
    ko_job *jobs = (ko_job *)calloc((size_t)T, sizeof(ko_job));
    pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
    pthread_barrier_t bar;
    volatile int nextBucket = 0;
    int rc = -1;
    if (km && km2 && rd && rd2 && rl && rs && kcount && koff && hist && cuts && ca && cu && ct && Mloc && jobs && th && bucketStart) {
        pthread_barrier_init(&bar, NULL, (unsigned)T);
        const double tStart = ko_now();
        for (int t = 0; t < T; ++t) {
            ko_job *j = &jobs[t];
            j->p = p; j->ix = ix; j->bases = bases; j->off = off; j->lut = lut; j->nThreads = T; j->tid = t;
            j->r0 = nReads * t / T; j->r1 = nReads * (t + 1) / T;
            j->kcount = kcount; j->km = km; j->km2 = km2; j->rd = rd; j->rd2 = rd2; j->nQ = nQ; j->koff = koff;
            j->hist = hist; j->bucketStart = bucketStart; j->nextBucket = &nextBucket; j->rs = rs; j->rl = rl;
            j->nReads = (uint64_t)nReads; j->ca = ca + cells * (size_t)t; j->cu = cu + cells * (size_t)t; j->ct = ct + cells * (size_t)t;
            j->M = Mloc; j->bar = &bar; j->cuts = cuts; j->zeroM = M ? 0 : 1; j->timing = getenv("KO_TIMING") != NULL; j->t0 = tStart;
            pthread_create(&th[t], NULL, ko_worker, j);
        }
        for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&bar);
        for (size_t i = 0; i < cells; ++i) {                       /* Compare.hpp:3445-3454 */
            double a = 0; uint64_t u = 0;
            for (int t = 0; t < T; ++t) { a += ca[cells * (size_t)t + i]; u += cu[cells * (size_t)t + i]; }
            if (countAll) countAll[i] = a;
            if (countUnique) countUnique[i] = u;
        }
        if (nQueries) *nQueries = nQ;
        rc = 0;
    }
    free(km); free(km2); free(rd); free(rd2); free(rl); free(rs); free(kcount); free(koff); free(hist); free(cuts);
    free(ca); free(cu); free(ct); if (!M) free(Mloc); free(jobs); free(th); free(bucketStart);
    return rc;
}

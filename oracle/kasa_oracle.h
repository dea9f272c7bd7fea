/*
 * kasa_oracle.h -- CPU restatement of kASA's `identify` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / reported baseline.  The product path (kasa_amd + libkasa_hip.so) never links or calls it.
 *
 * Every function cites the reference file:line (relative to the kASA v1.4.9 tree) it restates.
 * Parity pinning: tests/test_oracle_golden.py checks this restatement against outputs of the
 * reference's own shipped binary (binaries/kASA_linux, v1.4.9) captured by
 * tests/golden/make_fixtures.py -- see DESIGN.md "Oracle".
 */
#ifndef KASA_ORACLE_H
#define KASA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_RANGE_NONE UINT64_MAX

/* One source, two libraries (oracle/Makefile): libkasa_oracle.so for the 64-bit index (K = 12 letters,
 * uint64_t keys) and libkasa_oracle128.so, built with -DKO_WIDE, for the 128-bit index (K = 25 letters,
 * source/utils/uint128_t.hpp; a key is 16 bytes little endian: low word first, as packedLargePair stores
 * it, packedPairs.hpp:132-155). */
#ifdef KO_WIDE
typedef unsigned __int128 ko_key;
#else
typedef uint64_t ko_key;
#endif

typedef struct {
    int32_t K;        /* letters per packed k-mer of the index: 12 (64-bit index) or 25 (128-bit, KO_WIDE) */
    int32_t kHigh;    /* largest k evaluated  (-k <kHigh> <kLow>) */
    int32_t kLow;     /* smallest k evaluated */
    int32_t frames;   /* 3 (default) or 6 (--six); 1 (--one) */
    int32_t avxQuirk; /* 1: flush like the reference's scoreMatchAVX (n>3), i.e. like the shipped
                         -march=native binary; 0: scoreMatchNonAVX for every n (the parity target) */
    int32_t coverage; /* 1: also count countTotal (--coverage) */
    int32_t protein;  /* 1: the reads are amino-acid sequences (kASA.hpp:155-183, Read.hpp:60-81) */
    int32_t cmp64Quirk; /* KO_WIDE, ko_compare_sequential only: 1 = the comparator of compareWithDatabase sees just
                         the low 64 bits of its operands, as in the stock reference (a std::function declared with
                         uint64_t parameters, Compare.hpp:700-706); 0 = full width */
} ko_params;

typedef struct {
    const ko_key *kmer;       /* sorted by (kmer, taxid) */
    const uint32_t *tax;      /* DENSE taxon index (1..nTaxa-1), i.e. after the content-file map */
    uint64_t n;
    const uint32_t *triePrefix; /* _trie entries: 30-bit prefixes, ascending */
    const uint64_t *trieStart;  /* running sum of counts */
    const uint32_t *trieLenM1;  /* count - 1 */
    uint64_t nTrie;
    uint32_t nTaxa;           /* incl. index 0 = "non_unique" */
} ko_index;

/* kASA.hpp:621-667 -- the built-in codon table as a 366-entry LUT of 5-bit letter codes. */
void ko_codon_table(uint8_t lut[366]);

/* kASA::setCodonTable (kASA.hpp:579-615): the 64 codons of table `id` of an NCBI gc.prt file written over `lut`
 * ('*' -> '[').  Returns 1 when the table was found, 0 otherwise (lut untouched), -1 when the file cannot be read. */
int ko_codon_table_from_file(const char *path, const char *id, uint8_t lut[366]);

/* Read.hpp:633-654,612-630,36-57 -- padded length (incl. marker) and k-mer count of one raw read. */
int64_t ko_padded_len(int64_t rawLen, const ko_params *p);
int64_t ko_kmer_count(int64_t paddedLen, const ko_params *p);

/* Read.hpp:657-675,84-220,264-293 -- clean + pad + (revcomp) + translate + pack a batch of reads.
 * bases: concatenated raw read bytes, off[nReads+1].  Returns the number of k-mers; when outKmer is
 * NULL only counts.  Emission order = read order, forward strand then reverse complement. */
int64_t ko_encode_batch(const uint8_t *bases, const int64_t *off, int64_t nReads, const ko_params *p,
                        const uint8_t lut[366], ko_key *outKmer, uint32_t *outRead);

/* Compare.hpp:1077 -- sort by k-mer only (stable LSD radix here; ties do not matter). */
void ko_sort_queries(ko_key *kmer, uint32_t *read, uint64_t n);

/* Compare.hpp:3167-3178 (-e): drop a record equal in (k-mer, read id) to its predecessor.  The reference applies
 * std::unique to the output of an unstable sort by k-mer, so which duplicates end up adjacent is an accident of its
 * sort; after the stable sort here all of them are.  Identical whenever no other read shares the duplicated k-mer.
 * Returns the new count. */
uint64_t ko_unique_queries(ko_key *kmer, uint32_t *read, uint64_t n);

/* Compare.hpp:1098-1117 + Trie.hpp:494-520 -- prefix (kmer>>30) -> (start, len-1) or KO_RANGE_NONE. */
void ko_ranges(const ko_index *ix, const ko_params *p, const ko_key *kmer, uint64_t n,
               uint64_t *rangeStart, uint32_t *rangeLenM1);

/* Compare.hpp:678-1069 -- faithful sequential merge of the sorted queries against the index.
 * Tables are [lv * nTaxa + t] with lv = 0 for kHigh ... nK-1 for kLow (Compare.hpp:922).
 * M is the dense nReads x nTaxa float matrix (Utilities.hpp:592-636) or NULL (profile only). */
int ko_compare_sequential(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                          const uint32_t *qRead, const uint64_t *qRangeStart,
                          const uint32_t *qRangeLenM1, uint64_t nQ, uint64_t nReads,
                          double *countAll, uint64_t *countUnique, uint64_t *countTotal, float *M);

/* SURVEY.md section 0.1 -- the closed form the device implements (group by (range, k, prefix)),
 * evaluated with the same flush order.  Must equal ko_compare_sequential bit for bit. */
int ko_compare_closed_form(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                           const uint32_t *qRead, const uint64_t *qRangeStart,
                           const uint32_t *qRangeLenM1, uint64_t nQ, uint64_t nReads,
                           double *countAll, uint64_t *countUnique, uint64_t *countTotal, float *M);

/* --coherence.  The two comparison routines again, also reporting per sorted query the match length the reference's
 * setMatchLength leaves (Compare.hpp:847-848,882-884,912-914,948; sequential) resp. the deepest matched level (closed form):
 * matchLen / depth = u8[nQ], zeroed by the caller.  ko_coherence = Compare::postProcess (Compare.hpp:2607-2728) over the
 * batch's k-mers in (read, frame, position) order; returns 1 where the reference throws std::out_of_range. */
int ko_compare_sequential_ml(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                             const uint32_t *qRead, const uint64_t *qRangeStart,
                             const uint32_t *qRangeLenM1, uint64_t nQ, uint64_t nReads,
                             double *countAll, uint64_t *countUnique, uint64_t *countTotal, float *M, uint8_t *matchLen);
int ko_compare_closed_form_ml(const ko_params *p, const ko_index *ix, const ko_key *qKmer,
                              const uint32_t *qRead, const uint64_t *qRangeStart,
                              const uint32_t *qRangeLenM1, uint64_t nQ, uint64_t nReads,
                              double *countAll, uint64_t *countUnique, uint64_t *countTotal, float *M, uint8_t *depth);
int ko_coherence(const uint32_t *read, const uint32_t *pos, const uint8_t *frame, const uint8_t *len, uint64_t n,
                 int sixFrames, float *scores, uint64_t nReads, uint64_t *failIdx);

/* The whole batch with the reference's threading model (-n worker threads; Read.hpp:763-827, Compare.hpp:1123-1132,
 * :3263-3283, :3445-3454): translation per read chunk, parallel sort, lookup + score over range-aligned slices with
 * private count tables and the shared, unsynchronised score matrix.  For bench.py's cpu_baseline.  countAll /
 * countUnique / M may be NULL. */
int ko_identify_threaded(const ko_params *p, const ko_index *ix, const uint8_t *bases, const int64_t *off, int64_t nReads,
                         const uint8_t lut[366], int nThreads, double *countAll, uint64_t *countUnique, float *M, uint64_t *nQueries);

/* Compare.hpp:1452-1481 */
float ko_best_score(uint64_t readLen, const ko_params *p);
/* Compare.hpp:1510 -- relative score of one (read, taxon) cell. */
double ko_relative_score(float kmerScore, uint64_t freqAtKHigh, uint64_t readLen, const ko_params *p);
/* Compare.hpp:1634 -- error of one hit (float arithmetic, printed as double). */
double ko_error_score(float bestScore, float kmerScore);

/* Compare.hpp:392 -- w_k = k^2/625 as float. */
float ko_weight(int k);

#ifdef __cplusplus
}
#endif
#endif

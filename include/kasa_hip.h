/*
 * kasa_hip.h -- C ABI of the MI355X (gfx950) implementation of kASA's `identify` hot path.
 *
 * kASA has no plugin/FFI interface: the hot path is the set of calls that
 * Compare::CompareWithLib_partialSort (source/modes/Compare.hpp:2733) makes per batch.  This header
 * declares exactly those calls as a drop-in seam; each entry names the reference code it replaces.
 * Plain pointers and sizes only; every buffer passed in is owned by the caller and only borrowed for
 * the duration of the call; nothing returned outlives kasa_ctx_destroy / kasa_index_destroy.
 *
 * Errors: the reference throws std::runtime_error / bad_alloc and prints "ERROR: <what>"
 * (source/main.cpp:1717-1720).  Here every call returns a status (0 = ok) and never lets a C++
 * exception or HIP error cross the boundary; kasa_last_error() returns the message the host wrapper
 * rethrows as runtime_error (INTEGRATION.md shows the reference-side binding).
 *
 * Threading (Compare.hpp:3263-3283, main.cpp:1292-1326): a kasa_index is immutable after creation
 * and may be shared by any number of contexts / threads; a kasa_ctx (stream, batch buffers, profile
 * tables) is single-threaded.
 *
 * Table layout everywhere: [level * nTaxa + taxIdx], level 0 = kHigh ... nK-1 = kLow, taxIdx = dense
 * index of the content file (0 = "non_unique"), exactly Compare.hpp:922.
 */
#ifndef KASA_HIP_H
#define KASA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kasa_index kasa_index;
typedef struct kasa_ctx kasa_ctx;

enum {
    KASA_OK = 0,
    KASA_E_ARG = 1,       /* bad argument (the message says which) */
    KASA_E_HIP = 2,       /* a HIP runtime call failed */
    KASA_E_NOMEM = 3,     /* device or host allocation failed (reference: bad_alloc) */
    KASA_E_STATE = 4,     /* calls out of order (e.g. lookup before sort) */
    KASA_E_LIMIT = 5      /* a documented capacity limit was hit (message says which) */
};

/* stages of one batch, for kasa_ctx_stage_ms() */
enum {
    KASA_STAGE_ENCODE = 0,   /* Read.hpp:763-827  convertAndSort -> convertLinesTokMers_new */
    KASA_STAGE_SORT = 1,     /* Compare.hpp:1077/1130 sort by k-mer */
    KASA_STAGE_LOOKUP = 2,   /* Compare.hpp:1098-1117 + :803-829,:861-995: prefix table + search */
    KASA_STAGE_GROUP = 3,    /* Compare.hpp:917-955,1005-1041: per-level groups, taxon sets, flush order */
    KASA_STAGE_REGROUP = 4,  /* slots of the queries in read order when the encoder could not rank them (stable sort by read) */
    KASA_STAGE_SCORE = 5,    /* Compare.hpp:516-532 score accumulation in reference order */
    KASA_STAGE_COUNT = 6
};

/* Message of the last failed call on this thread ("" if none). */
const char *kasa_last_error(void);

/* Number of visible HIP devices (0 if none / no driver). */
int kasa_device_count(int *count);

/* ---- index residency: replaces Compare::ReadIndex::loadIndex / loadTrie / loadContentAndFrequencyFiles
 *      (source/modes/Compare.hpp:111-337).
 * records   : nRecords packed {u64 kmer, u32 taxid} entries of 12 bytes, sorted by (kmer, taxid) --
 *             the index file as it is on disk (may be an mmap of it); or, with recordBytes = 6, the
 *             "halved" index of shrink strategy 2, {u32 low 30 bits, u16 dense taxon index}
 *             (source/utils/packedPairs.hpp:100-105), which needs the trie arrays; or, with
 *             recordBytes = 20, the 128-bit index of `build --kH 25`: {u64 low, u64 high, u32 taxid}
 *             (packedLargePair, packedPairs.hpp:132-155; _info.txt carries the marker 128), 25 letters
 *             per k-mer, levels up to k = 25.
 * triePrefix/trieCount : the `_trie` file (source/modes/Trie.hpp:365-394): 30-bit prefixes ascending
 *             and their entry counts; may be NULL/0 -- the device derives its own two-level prefix
 *             table from the records and, when given, checks it against this one.
 * taxIds    : taxIds[i] = tax ID of dense index i as in the content file (taxIds[0] = 0), nTaxa
 *             entries (Compare.hpp:121-151).
 */
int kasa_index_create(int device, const void *records, uint64_t nRecords, int recordBytes,
                      const uint32_t *triePrefix, const uint64_t *trieCount, uint64_t nTrie,
                      const uint32_t *taxIds, uint32_t nTaxa, kasa_index **out);
void kasa_index_destroy(kasa_index *ix);
uint64_t kasa_index_size(const kasa_index *ix);
uint64_t kasa_index_device_bytes(const kasa_index *ix);

/* ---- context: replaces the kASA / Read / Compare constructors (source/kASA.hpp:276-305) and
 *      setCodonTable (source/kASA.hpp:579-615).
 * kHigh/kLow : -k <kHigh> <kLow>, at most 12 with a 64-bit index and 25 with a 128-bit one; frames: 3 (default), 6 (--six) or 1 (--one, Read.hpp:223-261);
 * codonLut: 366-byte table of 5-bit letter codes indexed like kASA.hpp:75, or NULL for the built-in
 * table (kASA.hpp:621-667).
 */
int kasa_ctx_create(const kasa_index *ix, int kHigh, int kLow, int frames, const uint8_t *codonLut,
                    kasa_ctx **out);
/* The built-in table (kASA.hpp:621-667) in the codonLut layout: the starting point for -a/--alphabet, which
 * overwrites the 64 codons of an NCBI gc.prt table in it (kASA::setCodonTable, kASA.hpp:579-615).  Host only. */
int kasa_builtin_codon_table(uint8_t *lut366);
void kasa_ctx_destroy(kasa_ctx *ctx);

/* The next batches hold amino-acid sequences instead of DNA (what kASA::detectAlphabet decides per
 * input file, kASA.hpp:155-183): letters are taken as they are ('*' -> '[', code = char & 31), padding
 * and marker are '^', a read of L letters gives L-K+1 k-mers (Read.hpp:36-41,60-81,636-640,663-667,
 * 1069-1073); --six is switched off (kASA.hpp:181).  protein = 0 returns to DNA. */
int kasa_ctx_set_protein(kasa_ctx *ctx, int protein);

/* ---- one batch ------------------------------------------------------------------------------- */

/* Hand the raw reads of one batch to the device: concatenated bases and offsets[nReads+1].
 * Replaces the vLines the reference keeps per batch (Read.hpp:612-630).  Cleaning (non-ACGT -> Z),
 * padding and the X marker (Read.hpp:633-675,1068-1078) happen on the device.  One copy only; `bases` may also point into
 * device memory (a host that keeps the reads of a whole file resident, as bench.py does for its multi-batch steps);
 * `offsets` is host memory. */
int kasa_batch_upload(kasa_ctx *ctx, const uint8_t *bases, const int64_t *offsets, int64_t nReads);

/* The same for a host that keeps the reads of a whole file resident in HBM (bench.py's steps at every N, a pipeline that
 * parses on the device): `basesDev` AND `offsetsDev` (int64[nReads+1], relative to basesDev) lie in device memory.  Nothing
 * crosses PCIe and the bases are NOT copied: the context reads them in place, so they must stay valid and unchanged until
 * the batch's last call (kasa_batch_coherence re-reads them).  What is left of the upload is the geometry of Read.hpp:36-57,
 * 633-675 on the device (k-mers per read, running sums) and one 8-byte read-back of the batch's k-mer count. */
int kasa_batch_upload_device(kasa_ctx *ctx, const uint8_t *basesDev, const int64_t *offsetsDev, int64_t nReads);

/* The same for reads made of several sequences: paired-end input (-1/-2; Read::readFastqa_pairedEnd,
 * Read.hpp:834-1049) hands both mates of a pair on as two entries of vLines with ONE read id, so their
 * k-mers score into the same row and none spans the junction.  offsets[nSegments+1] delimit the
 * sequences, segmentRead[s] (ascending, < nReads) names the read of sequence s. */
int kasa_batch_upload_segments(kasa_ctx *ctx, const uint8_t *bases, const int64_t *offsets, int64_t nSegments,
                               const uint32_t *segmentRead, int64_t nReads);

/* Read::convertAndSort (Read.hpp:763-827): every read -> packed k-mers + read id, device resident. */
int kasa_batch_encode(kasa_ctx *ctx, uint64_t *nKmers);

/* Compare::sortInputAndCheckInvalidkMers_sta (Compare.hpp:1074-1260): sort by k-mer and find each
 * query's place in the index.  unique != 0 is -e/--unique (Compare.hpp:3167-3178): records equal in
 * (k-mer, read id) are kept once.  The reference applies std::unique to an unstable sort, so duplicates
 * it happens not to place side by side survive there; here all of them are dropped. */
int kasa_batch_sort_and_range(kasa_ctx *ctx, int unique);

/* Compare::compareWithDatabase + scoreMatchForReadIDsAndTaxIDs (Compare.hpp:678-1069,516-673):
 * adds this batch into the context's profile tables; with wantPerRead also produces the non-zero
 * cells of the reads x taxa score matrix (Utilities.hpp:592-636) as a CSR. */
int kasa_batch_lookup_score(kasa_ctx *ctx, int wantPerRead, int coverage);

/* The two halves of kasa_batch_lookup_score as separate steps, with the event records in between exposed, for an index
 * that is range-partitioned over devices (C5; the reference's loadIndex has no such mode, the seam is ours):
 *   kasa_batch_group          one record per sorted query, in sorted order: 8 (up to 8 levels) or 16 (up to 25 levels)
 *                             32-bit words {position, last flush position, deepest level | flush order of its events |
 *                             flags, number of taxon segments | sizes of the taxon sets per level (narrow records),
 *                             [wide: the flush order, 4 words], up to 4 (8) segments -- or 3 (7) and the pool offset of a
 *                             longer list: pool block = {number of segments, [4 words of exact set sizes when the
 *                             record's "crowded" flag is set], the further segments}};
 *                             a segment = taxon | first level << 22 | last level << 27, so an index may name at most
 *                             2^22 taxa (layout: kasa_amd/csrc/kasa_hip.hip "event records", DESIGN.md sections 3-4).
 *                             Records and pools are opaque to a caller that only moves them; they are NOT a stable
 *                             format across versions of this library;
 *   kasa_batch_records_*      size (in 32-bit words) / download of records and pool; import hands records in sorted order
 *                             to a context whose batch is sorted, which files them by read;
 *   kasa_batch_score          replays the records per read (scores, profile) exactly as kasa_batch_lookup_score does. */
int kasa_batch_group(kasa_ctx *ctx, int coverage);
/* kasa_batch_group with the records written straight into the caller's device buffer (room for
 * number of queries x record words u32): the partition worker groups a slice into the tensor its collective sends, no copy
 * in between.  kasa_batch_records_device then reports that pointer; the pool stays the context's. */
int kasa_batch_group_to(kasa_ctx *ctx, int coverage, uint32_t *recordsOutDev);
int kasa_batch_score(kasa_ctx *ctx, int wantPerRead);
int kasa_batch_records_size(kasa_ctx *ctx, uint64_t *nRecordWords, uint64_t *nPoolWords);
int kasa_batch_records_fetch(kasa_ctx *ctx, uint32_t *records, uint32_t *pool);
int kasa_batch_records_import(kasa_ctx *ctx, const uint32_t *records, uint64_t nRecordWords, const uint32_t *pool, uint64_t nPoolWords);

/* The same exchange without the host: slices of the sorted queries and the records made from
 * them are handed over as DEVICE pointers, so the caller's collective (RCCL all_to_all over xGMI) moves them from HBM to
 * HBM.  Pointers returned by the *_device getters point into the context's own buffers and stay valid until its next
 * batch call; pointers passed in may live on the context's device or on a peer whose memory it can address.
 *   kasa_batch_queries_device         the sorted k-mers of the batch (8 or 16 bytes each) and their number
 *   kasa_batch_slice_starts           starts[j] = first sorted query whose 30-bit prefix is >= cuts[j] (starts[0] = 0,
 *                                     starts[nParts] = number of queries): slice j goes to the owner of partition j
 *   kasa_batch_set_sorted_device      installs a slice of sorted k-mers as a batch of its own and looks it up
 *                                     (kasa_batch_set_queries + kasa_batch_sort_and_range without the sort);
 *                                     kasa_batch_group then makes its records
 *   kasa_batch_records_device         records (sorted order) and pool of the grouped slice
 *   kasa_batch_records_import_device  the slices' records, in partition order, become the batch's records: positions
 *                                     move by the slice starts and pool offsets by the pool bases on the device
 *                                     (what kasa_amd/partition.py:assemble_records does for the host path) */
int kasa_batch_queries_device(kasa_ctx *ctx, const void **kmers, uint64_t *n);
int kasa_batch_slice_starts(kasa_ctx *ctx, const uint64_t *cuts, uint32_t nParts, uint64_t *starts);
int kasa_batch_set_sorted_device(kasa_ctx *ctx, const void *kmersDev, uint64_t n);
int kasa_batch_records_device(kasa_ctx *ctx, const uint32_t **records, uint64_t *nRecordWords, const uint32_t **pool, uint64_t *nPoolWords);
int kasa_batch_records_import_device(kasa_ctx *ctx, uint32_t nParts, const uint32_t *const *records, const uint64_t *nRecordWords,
                                     const uint32_t *const *pool, const uint64_t *nPoolWords);
/* The staging buffer kasa_batch_records_import_device files the records from, with room for nRecordWords words: a caller that
 * receives the slices' records THERE (back to back, in partition order) and names those places as `records[j]` spares the
 * batch a second copy of its records (32 or 64 bytes per query); they are then shifted in place. */
int kasa_batch_records_inbox(kasa_ctx *ctx, uint64_t nRecordWords, uint32_t **records);
/* The records of an exported slice PACKED for the wire (the return leg of the partitioned exchange; ours -- the reference streams
 * its index from disk and exchanges nothing): one byte of classes per four records (2 bits each: unmatched / words [1..4] /
 * [1..6] / [1..7] of a 32-byte record, [1..9] / [1..11] / [1..15] of a 64-byte one, by its number of segments), padded to 16
 * bytes, then those words back to back; word [0] -- the query's place in the slice -- is implied.  SURVEY 8(e) sizes the
 * exchange at 12-20 bytes per query.  pack_size (one pass over the records: the sizes) then pack, on the context that grouped
 * the slice (records = what kasa_batch_group_to / kasa_batch_records_device gave); unpack on the read owner, into the place
 * the whole records would have been received at (kasa_batch_records_inbox), before kasa_batch_records_import_device. */
int kasa_batch_records_pack_size(kasa_ctx *ctx, const uint32_t *recordsDev, uint64_t nQueries, uint64_t *nBytes);
int kasa_batch_records_pack(kasa_ctx *ctx, const uint32_t *recordsDev, uint64_t nQueries, void *outDev, uint64_t capBytes);
int kasa_batch_records_unpack(kasa_ctx *ctx, const void *packedDev, uint64_t nBytes, uint64_t nQueries, uint32_t *recordsOutDev);


/* CSR of the batch: readOffsets[nReads+1]; per read taxIdx ascending with score > 0 -- the cells
 * scoringFunc scans (Compare.hpp:1501-1522). */
int kasa_batch_scores_size(kasa_ctx *ctx, uint64_t *nnz);
int kasa_batch_scores_fetch(kasa_ctx *ctx, uint64_t *readOffsets, uint32_t *taxIdx, float *score);

/* --coherence (Compare::postProcess, Compare.hpp:2607-2728, fed by setMatchLength at :847-848,882-884,912-914,948; printed at
 * :1662-1665 and used by --filter at :1602-1606).  Per read the reference's coherence score of
 * the batch that was last sorted (kasa_batch_sort_and_range): the k-mers in the order the reader emitted them (read, strand,
 * window), match length = the deepest matched level of the k-mer, clusters of overlapping matches walked with the
 * reference's own state machine -- including what it credits to reads without k-mers and what it skips after a strand
 * switch.  scores = float[nReads] (host).  *throwsAt = ~0, or the index at which the reference's walk runs off the end of
 * its vector (std::out_of_range from vector::at, Compare.hpp:2667: with --six, a batch whose last strand switch finds no
 * further match): then the reference ends with "vector::_M_range_check: __n (which is N) >= this->size() (which is N)",
 * N = *throwsAt, and the host should do the same.  Single-end batches without -e only (the reference's results for the
 * others depend on its unstable sorts): KASA_E_ARG otherwise. */
int kasa_batch_coherence(kasa_ctx *ctx, float *scores, uint64_t *throwsAt);

/* Ranking on the device (Compare::scoringFunc, Compare.hpp:1495-1594 and the printing loops
 * :1721-1754): instead of the whole CSR only what the per-read file can print leaves the device.
 *   den        [nClasses][nTaxa] doubles, row c = 1 + log2(freq[t] * double(uint32(len_c - 3K + 1))) for the c-th distinct
 *              read length of the batch (len_c - K + 1 for protein input), computed by the host with libm exactly as
 *              Compare.hpp:1506-1511 does; readClass[r] = the row of read r
 *   threshold  -t (relative scores below it are dropped), beasts = -b
 * Per read the hits are taken in the order (relative score descending, taxon ascending) for as long as the TSV or the
 * JSON / JSONL / Kraken writer would print another one.  kasa_batch_rank_fetch delivers
 *   meta[4 r .. 4 r + 3] = { first entry, number of entries | flag << 31, float bits of the largest k-mer score among
 *                            the hits, number of hits }
 *   entries[i]           = { uint32 taxIdx, float kmerScore, double relativeScore }   (16 bytes)
 * (nEntries = length of the entries array; reads own disjoint ranges of it, handed out in slabs, so some entries
 * between them are unused)
 * A writer that runs the reference's loops over these entries (as if they were all hits, with the delivered maximum)
 * prints what it would print from the full row.  Where more than 16 hits meet a tie in the relative score inside the
 * printed prefix, the order is std::sort's own (not stable, but deterministic): those reads are ranked by a second kernel
 * that walks libstdc++'s introsort over the hits (kasa_amd/csrc/stdsort_order.h).  flag: the read could not be ranked on
 * the device (std::sort would have switched to its heap sort, or the row has 65 536 or more cells): it gets no entries
 * and the host ranks it from its full row (kasa_batch_scores_fetch).  nFlagged counts such reads.
 * The call may use the batch's event records as scratch: kasa_batch_group has to run again before another kasa_batch_score
 * of the same batch (the scores themselves, kasa_batch_scores_fetch, stay valid). */
int kasa_batch_rank(kasa_ctx *ctx, const double *den, uint32_t nClasses, const uint32_t *readClass, float threshold, uint32_t beasts,
                    uint64_t *nEntries, uint32_t *nFlagged);
int kasa_batch_rank_fetch(kasa_ctx *ctx, uint32_t *meta, void *entries);

/* The per-read file's text written on the device (the printing loops of Compare::scoringFunc,
 * Compare.hpp:1526-1872, with the reference's number formats utils/iToStr.hpp:35-114 and utils/dToStr.h:427-456): after
 * kasa_batch_rank the hits never leave the device -- a kernel writes every read's TSV / JSON / JSON-lines / Kraken bytes
 * into one buffer (read order, offsets by a prefix sum of the sizes), and what crosses PCIe is the next piece of the file.
 *   kasa_ctx_set_taxa_text  once per context: taxIds[nTaxa] and the names of the content file (entry 0 = "non_unique"),
 *                           back to back, names[nameOff[t] .. nameOff[t+1])
 *   kasa_batch_text         the text of the batch last ranked.  KASA_E_STATE when kasa_batch_rank left reads to the host
 *                           (nFlagged > 0: the host then writes the whole batch from kasa_batch_rank_fetch +
 *                           kasa_batch_scores_fetch as before)
 *   kasa_batch_text_fetch   text[nBytes] (NULL: the text stays where it is, see kasa_batch_text_fetch_range); readOffsets[nReads + 1]
 *                           (may be NULL): where each read's text starts;
 *                           contaminated[nReads] (may be NULL): 1 = --filter's rule holds for the read (Compare.hpp:1597-1606:
 *                           within errorThreshold of the perfect score, or coherence >= coherenceThreshold) */
enum { KASA_TEXT_TSV = 0, KASA_TEXT_JSON = 1, KASA_TEXT_JSONL = 2, KASA_TEXT_KRAKEN = 3 };
typedef struct kasa_text_params {
    int format;                    /* KASA_TEXT_* */
    uint32_t beasts;               /* -b */
    uint64_t firstRead;            /* number of the batch's first read in its file ("Read number") */
    const char *readNames;         /* host: the specifiers as printed, back to back */
    const uint64_t *readNameOff;   /* host: [nReads + 1] */
    const uint32_t *readLen;       /* host: "Length" of every read */
    const float *bestScore;        /* host: [nClasses] the perfect score of a read of class c (Compare.hpp:1452-1481); classes =
                                      the readClass given to kasa_batch_rank */
    uint32_t nClasses;
    int coherence;                 /* print the coherence of the read (the scores of kasa_batch_coherence on this batch) */
    double errorThreshold;         /* --errorThreshold, as a double */
    float coherenceThreshold;      /* --coherenceThreshold */
} kasa_text_params;
int kasa_ctx_set_taxa_text(kasa_ctx *ctx, const uint32_t *taxIds, const char *names, const uint64_t *nameOff);
int kasa_batch_text(kasa_ctx *ctx, const kasa_text_params *params, uint64_t *nBytes);
int kasa_batch_text_fetch(kasa_ctx *ctx, char *text, uint64_t *readOffsets, uint8_t *contaminated);
/* text[offset .. offset + nBytes) of the same text: a host that moves it through a few small page-locked buffers (making one
 * of 5.5 GB for the text of 10 M reads takes seconds) writes one piece to the file while the next one arrives. */
int kasa_batch_text_fetch_range(kasa_ctx *ctx, char *text, uint64_t offset, uint64_t nBytes);
/* Test tap: the reference's double -> text (dToStr.h) as the device writes it; out = 32 bytes per value, zero-terminated. */
int kasa_text_dtoa(int device, const double *values, uint32_t n, char *out);

/* Page-locked host memory for buffers that cross PCIe (reads in, ranked hits or CSR out).  NULL when it cannot be had. */
void *kasa_host_alloc(size_t bytes);
void kasa_host_free(void *p);
/* Plain device memory for a host that keeps its inputs resident in HBM (kasa_batch_upload_device) and holds no other GPU
 * library: allocate, fill from host memory, free.  Ours (the reference has no device): bench.py's one-GPU run keeps its reads
 * in these, so that process never imports torch and the library runs on the HIP runtime it was built for. */
int kasa_device_alloc(int device, size_t bytes, void **out);
int kasa_device_free(int device, void *p);
int kasa_device_write(int device, void *dst, const void *src, size_t bytes);
int kasa_device_read(int device, void *dst, const void *src, size_t bytes);
/* Binds the CALLING host thread to a device.  A fresh thread stands on device 0, and kasa_host_alloc page-locks for the
 * thread's current device: a helper thread that prepares a worker's buffers calls this first (the kasa_ctx_* / kasa_batch_*
 * calls select their context's device themselves). */
int kasa_thread_device(int device);

/* ---- profile tables: vCount_all / vCount_unique / vCount_total (Compare.hpp:2830-2839) ---------- */
int kasa_profile_reset(kasa_ctx *ctx);
/* dst += src, src = 0.  The profile of a batch is made where its queries are GROUPED (kasa_batch_lookup_score, kasa_batch_group):
 * a sum over (group, taxon) of the group's hits (Compare.hpp:922-925), from the sorted queries -- with 32-byte records (up to 8
 * levels) kasa_batch_score adds nothing to the tables; with 64-byte records (9-25 levels, the 128-bit index's default k range)
 * the profile still comes from the per-read side, i.e. from kasa_batch_score.  A context that groups slices for another one
 * (the partition worker of a range-partitioned index) hands its tables on with this call.  Same device, k range and taxa. */
int kasa_profile_absorb(kasa_ctx *dst, kasa_ctx *src);
/* countAll as double (exact 64.64 fixed-point sums rounded once), countUnique, countTotal; any may
 * be NULL.  nK * nTaxa entries each. */
int kasa_profile_fetch(kasa_ctx *ctx, double *countAll, uint64_t *countUnique, uint64_t *countTotal);
/* The same tables as 6 integer limbs per cell {unique, total, all[0..3] (32 bits each, little
 * endian)} so that ranks can be summed with a plain integer reduce (the thread reduce of
 * Compare.hpp:3445-3454 across GPUs); import REPLACES the context's tables by the limbs handed in
 * (the sum over all ranks, after the reduce). */
int kasa_profile_export_limbs(kasa_ctx *ctx, uint64_t *limbs);
int kasa_profile_import_limbs(kasa_ctx *ctx, const uint64_t *limbs);
/* The whole reduce on the device: limbs packed in HBM, ONE ncclAllReduce (u64, sum; exact and order-independent) on
 * the context's stream over the communicator handed in (an ncclComm_t of RCCL, passed as void * so that this header
 * needs no RCCL include), carries folded back.  Afterwards every rank's tables hold the global sums.  Collective:
 * every rank of the communicator must call it. */
int kasa_profile_allreduce(kasa_ctx *ctx, void *rcclComm);
/* RCCL is bound at run time to the copy the process already holds -- the one that made `rcclComm` (the host's own link,
 * a torch wheel's bundled librccl, a ctypes load); librccl.so.1 is loaded only when the process has none.  What this
 * library was built with and what it runs on: HIP versions as HIP_VERSION (major * 10^7 + minor * 10^5 + patch), RCCL's as
 * NCCL_VERSION_CODE; *rcclRuntime = 0 when no RCCL is in the process.  Any pointer may be NULL.  (No reference
 * counterpart: kASA is one statically linked binary; this is what lets a host that shares its process with another ROCm
 * stack see a mismatch instead of running on it unawares.) */
int kasa_runtime_versions(int *hipBuilt, int *hipRuntime, int *hipDriver, int *rcclBuilt, int *rcclRuntime);

/* ---- measurement + test taps ------------------------------------------------------------------ */
/* HIP-event time (ms) and launch count of a stage, accumulated since the last reset. */
int kasa_ctx_stage_ms(kasa_ctx *ctx, int stage, double *ms, uint64_t *launches);
int kasa_ctx_stage_reset(kasa_ctx *ctx);
/* HIP-event time of single kernels alone (accumulated since the last kasa_ctx_stage_reset), for the roofline lines. */
enum {
    KASA_KERNEL_LOOKUP = 0,       /* lookup_tile_kernel: the sorted-index lookup */
    KASA_KERNEL_GROUP = 1,        /* group_kernel: flush order + taxon segments, one record per query */
    KASA_KERNEL_SCORE_MAIN = 2,   /* score_main_kernel: the float chains of a read's register taxa */
    KASA_KERNEL_SCORE_OTHER = 3,  /* score_other_kernel: staging records of all other taxa */
    KASA_KERNEL_ROW_MERGE = 4,    /* row_merge_*: staging rows -> final {taxon, score} rows + profile keys */
    KASA_KERNEL_SCORE_GENERAL = 5,/* flush_positions_kernel + score_kernel (all its passes): the reads the fast kernels hand over */
    KASA_KERNEL_PROFILE_TABLES = 6,/* profile_(group_)table_kernel + the sort and reduce of the keys it leaves over */
    KASA_KERNEL_ROW_COPY = 7,     /* row lengths -> CSR offsets (scan) + row_copy_kernel */
    KASA_KERNEL_SORT_PASSES = 8,  /* kasa_radix: hist_kernel + the radix passes of the query sort */
    KASA_KERNEL_BUCKET_RANK = 9,  /* bucket_rank*_kernel: the query sort's last step */
    KASA_KERNEL_SCORE_DENSE = 10, /* score_dense_kernel: reads with long rows that keep the fast kernels' order rule (crowded indices) */
    KASA_KERNEL_SCORE_REPLAY = 11,/* kasa_replay.h: very long reads -- events sorted by (read, taxon, flush position, level), float chains per (read, taxon) */
    KASA_KERNEL_COUNT = 12
};
int kasa_ctx_kernel_ms(kasa_ctx *ctx, int kernel, double *ms, uint64_t *launches);
/* Of the last batch: {queries, staging records, profile keys, pool words, reads on the general kernel, of those on its
 * second pass, non-zero score cells, 1 if the encoder ranked the reads' k-mers (no slot fix-up)}. */
int kasa_ctx_batch_stats(kasa_ctx *ctx, uint64_t *stats8);
/* Of the last batch's group stage: the tiles of 1024 sorted queries it had, and how many of them the dense-leader kernel
 * (group2_kernel) left to the general one -- tiles with long taxon lists (Compare.hpp:396-441, a conserved k-mer) or walks
 * beyond its staged index span; 0 when the general kernel ran alone. */
int kasa_ctx_group_tiles(kasa_ctx *ctx, uint32_t *tiles, uint32_t *listed);
/* Of the last batch's score stage: the reads score_dense_kernel scored -- reads with long rows (many taxa per k-mer:
 * Compare.hpp:396-441) whose groups all close before the read's next matched query, replayed query by query from the row in
 * LDS; the others of the fast kernels' leftovers are the general kernel's (kasa_ctx_counters). */
int kasa_ctx_dense_reads(kasa_ctx *ctx, uint32_t *denseReads);
/* Number of query records the batch holds right now: the k-mer count of kasa_batch_encode, less the
 * duplicates once kasa_batch_sort_and_range ran with unique != 0. */
int kasa_batch_query_count(kasa_ctx *ctx, uint64_t *n);

/* Queries of the batch after encode / after sort (k-mer, read id); for parity tests.  A k-mer is one
 * uint64_t with a 64-bit index, two (low word, high word) with a 128-bit index. */
int kasa_batch_fetch_queries(kasa_ctx *ctx, void *kmers, uint32_t *reads, uint64_t n);
/* Test tap: install (k-mer, read id) queries directly instead of upload + encode (any order). */
int kasa_batch_set_queries(kasa_ctx *ctx, const void *kmers, const uint32_t *reads, uint64_t n, int64_t nReads);
/* Per sorted query: deepest matched level k (0 = none) and an index position sharing that prefix. */
int kasa_batch_fetch_lookup(kasa_ctx *ctx, uint8_t *depth, uint32_t *indexPos, uint64_t n);
int kasa_ctx_device_bytes(kasa_ctx *ctx, uint64_t *bytes);
/* Sizing a batch (the reference sizes its batches by -m, Compare.hpp:2803-2876): free / total HBM of a device, and the
 * device bytes one query k-mer costs while its batch is in flight (all stages resident). */
int kasa_device_memory(int device, uint64_t *freeBytes, uint64_t *totalBytes);
uint64_t kasa_batch_bytes_per_query(const kasa_ctx *ctx);

/* ---- where the reference cuts its batches (host arithmetic, no device work) ---------------------------------------
 * Per-read scores are float sums whose order depends on the reads that share a batch, so byte-identical per-read files
 * need the reference's batch boundaries: `kASA identify -m <GB>` turns -m into a byte budget
 * (main.cpp:438-447,590-592,1054-1060; Compare.hpp:111-160,182-328,2803-2818,3129-3132) that every read consumes
 * (Read.hpp:612-630,1147,1165-1195).  A host calls kasa_refbatch_budget once, kasa_refbatch_sequence_cost / _read_overhead per read and
 * kasa_refbatch_cut per batch; the device takes whatever batch it is handed. */
typedef struct {
    int64_t memoryGiB;            /* -m (0: the default, 5) */
    int threads;                  /* -n */
    int ram;                      /* -r */
    int kHigh, kLow;
    int recordBytes;              /* of the index FILE: 12, 20 (128-bit) or 6 (halved) */
    uint64_t nRecords;
    const uint32_t *triePrefix;   /* the `_trie` file's prefixes, ascending */
    uint64_t nTrie;
    const uint32_t *taxIds;       /* as for kasa_index_create: entry 0 = 0 */
    uint32_t nTaxa;
    uint64_t nameBytes;           /* bytes of all names of the content file after removing ',' (entry 0 not counted) */
    int identifyMultiple;         /* main.cpp:1118-1334 */
} kasa_refbatch_params;
int kasa_refbatch_budget(const kasa_refbatch_params *p, int64_t *budget);
/* One sequence (a read, or one mate of a pair): K = 12 or 25 (index type), mode 0 = DNA in 3/6 frames, 1 = --one,
 * 2 = amino acids; strands = 2 with --six.  The read as a whole, when per-read results are kept (-q / --filter):
 * nameLen = length of the specifier the reference stores (header without its first character + one space; both mates').
 * coherence: --coherence widens the k-mer records (InputType::ppTuple, MetaHeader.h:172) and adds a float per read
 * (Read.hpp:1191-1193). */
int64_t kasa_refbatch_sequence_cost(int K, int kLow, int mode, int strands, int64_t rawLen, int coherence);
int64_t kasa_refbatch_read_overhead(int64_t nameLen, uint32_t nTaxa, int coherence);
/* Reads of the next batch given the costs of the reads still to come. */
uint64_t kasa_refbatch_cut(int64_t budget, int firstBatch, const int64_t *cost, uint64_t nReads);

/* Device buffers for a batch of about nQueries k-mers out of nBases bases, before the batch is there: allocation takes
 * 25-90 ms per GB on this platform (seconds for a 10 M-read batch), time a host has while it parses its input.  Sizes only --
 * a batch that needs more gets more when it arrives.  Not while a batch of this context is in flight. */
int kasa_ctx_reserve(kasa_ctx *ctx, uint64_t nQueries, uint64_t nBases, int wantPerRead);

/* Of the last batch: reads scored by the general kernel (score_kernel) instead of the lane-per-read one, and how many of
 * those needed its second pass (full pending window / direct profile adds).  Diagnostics for tests and bench.py. */
int kasa_ctx_counters(kasa_ctx *ctx, uint32_t *generalReads, uint32_t *secondPassReads);
/* ... and how many of those kept more groups pending than the second pass's window holds (4096) and were ordered through a
 * window in device memory (narrow records; 64-byte records return KASA_E_LIMIT there).  Ours: the reference's flush is a
 * hash map per thread and has no such window (Compare.hpp:917-955). */
int kasa_ctx_third_pass_reads(kasa_ctx *ctx, uint32_t *thirdPassReads);
/* ... and the reads of that list that were long enough (16384 k-mers and more: a contig, a chromosome read in pieces under one
 * read id) to be replayed from SORTED EVENTS instead -- every query of theirs turned into events {read, taxon, flush position,
 * level} by all wavefronts, one radix sort, one float32 chain per (read, taxon) -- and the events that took.  The reference
 * streams such a sequence at merge speed (Compare.hpp:747-1043; pieces: Read.hpp:437-443,678-695); one wavefront per read
 * does not.  Both record widths (64-byte records: the replay also adds the events to the profile tables). */
int kasa_ctx_replay_stats(kasa_ctx *ctx, uint32_t *reads, uint64_t *events);
/* Of the tiles kasa_ctx_group_tiles reports as listed by the dense-leader kernel: those that its second launch -- the same
 * kernel with a park buffer four times as large, for tiles that only parked too many segments (a heavy 7-letter group) --
 * listed AGAIN and left to the cooperative kernel (long taxon lists, walks beyond the staged index span).  Ours. */
int kasa_ctx_group_second_chance(kasa_ctx *ctx, uint32_t *listedAgain);
/* How the buffer of 32-byte event records was chosen.  The rate at which an MI355X takes random 32-byte stores is a property
 * of the physical memory behind a buffer (21 to 28 G records/s, tools/place_probe.hip), and the group stage ends in one such
 * store per query: a buffer of 8 GB and more (KASA_PLACE_MIN_MB) is taken from up to three candidates (KASA_PLACE_TRIES),
 * each timed with a few milliseconds of scattered stores when it is allocated.  candidates = 0: allocated plainly (a small
 * buffer, 64-byte records); keptRate and rates4[0..3] in G records/s.  rates4 may be NULL.  Ours: the reference has no
 * device memory (its record of a k-mer's hits is the host vector of Compare.hpp:1430-1467). */
int kasa_ctx_record_placement(kasa_ctx *ctx, uint32_t *candidates, float *keptRate, float *rates4);

/* Test tap: forceSlowScore >= 0 is a bit set: bit 0 = every read takes the general (wavefront-per-read)
 * score kernel, bit 1 = per-query index search instead of the streamed-tile lookup, bit 2 = sorting row
 * merge instead of the bitmap one, bit 3 = the encoder does not rank the k-mers of a read (slots come from
 * a stable sort by read id, as for long or paired reads), bit 4 = the profile keys are sorted and reduced even
 * when the LDS counting table would fit, bit 5 = score_other_kernel (one lane per query) instead of
 * score_other_flat_kernel (work items = segments) for 32-byte records, bit 6 = the library's radix sort over all
 * key bits instead of 4 (5) passes + bucket_rank_kernel, bit 7 = long sort buckets are not sorted one by one but by the
 * library over all bits (the path for inputs with too many or too long ones), bit 8 = kasa_batch_rank leaves reads with
 * tied hits to the host instead of ranking them with std::sort's order on the device, bit 9 = the library's radix passes
 * over the top 32 (40) key bits instead of the hand-written ones (kasa_amd/csrc/kasa_radix.h), bit 10 = score_main_kernel with the
 * next line of records prefetched into registers (fewer resident wavefronts), bit 11 = no followers (every query walks the
 * index itself), bit 12 = 64-byte records without the index span in LDS, bit 13 = the general score kernel in its
 * lane-owns-its-cells form (what indices beyond 16 384 taxa take) instead of the row in LDS, bit 14 = the general kernel's
 * first pass hands every read to its second pass, bits 15 / 16 = timing taps of group_kernel (no profile keys at all; keys
 * counted but not stored: the profile is wrong), bit 17 = never the cooperative form of group_kernel (long taxon lists lane
 * by lane; level sizes beyond 255 wrap), bit 18 = the cooperative form from the first batch on, bit 19 = the radix passes of the
 * query sort look back over the earlier tiles before they order their keys in LDS (the form of rounds 2-3), bit 20 =
 * rank_exact_kernel in its largest form for every read (no classes by hit count), bit 21 = the other split of the query
 * sort (64-bit keys: five passes over 40 bits + buckets of 20 bits' worth; 128-bit keys: four passes over 32 bits + buckets
 * of 93), bit 22 = the bucket pass of 64-bit keys by the kernel for any key width, bit 23 = the general kernel's second pass
 * hands every read to its third (pending window in device memory; narrow records), bit 27 = narrow records are stored as
 * whole 64-byte cells (group2_kernel writes the record and 32 bytes of zeros by lane quads: a layout option, also
 * KASA_WIDE_CELLS=1), bit 29 = never the sorted-event replay of very long reads (kasa_ctx_replay_stats), bit 30 = that replay for
 * every read of the general kernel's list, whatever its length; lastSlowReads (may be
 * NULL) receives how many reads of the last batch took the general score kernel. */
int kasa_ctx_debug(kasa_ctx *ctx, int forceSlowScore, uint32_t *lastSlowReads);
int kasa_ctx_synchronize(kasa_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif

// kasa_refbatch.cpp -- where the reference cuts its batches (host arithmetic only; no device code).
//
// Per-read scores are float sums whose order depends on which reads share a batch, so a
// host that wants byte-identical per-read output has to cut its batches exactly where `kASA identify -m <GB>` does.
// The reference derives a byte budget from -m and lets every read consume part of it; these functions restate that
// arithmetic.  They decide only WHERE to cut -- the device takes any batch.
//
//   budget : source/main.cpp:438-447,590-592 (-m in GiB, default 5), :1054-1060 (trie, content, index),
//            source/modes/Compare.hpp:111-160 (content + frequency memory), :182-328 (index in RAM or STXXL caches),
//            :2803-2818 (average usage), :3129-3132 (0.1 % once after the first batch)
//   trie   : source/modes/Trie.hpp:74-99 (256 B per inner node, sizeof(Leaf5) per 5-letter leaf)
//   reads  : source/modes/Read.hpp:612-630 (k-mers and padded text of a read), :1147 (stop at <= 100 MiB),
//            :1165-1195 (name, length, score row)
#include <cstdint>
#include <cstring>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/kasa_hip.h"

namespace {
constexpr int64_t GIB = 1024ll * 1024ll * 1024ll;

// Trie<intType>::LoadFromStxxlVec -> Node::IncreaseIndex (Trie.hpp:74-99): every distinct 1..4-letter prefix below the
// root allocates a Node (counted as 256 bytes), every distinct 5-letter prefix a Leaf5 (32 x u64 + 32 x u32).
int64_t trie_bytes(const uint32_t *prefix, uint64_t n)
{
    int64_t bytes = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t p = prefix[i];
        // prefixes arrive sorted: a node is new iff its letters differ from the previous prefix's
        int shared = 0;                                             // leading letters shared with the previous prefix
        if (i > 0) {
            const uint32_t x = p ^ prefix[i - 1];
            shared = 6;
            for (int l = 0; l < 6; ++l)
                if ((x >> (5 * (5 - l))) & 31u) { shared = l; break; }
        }
        for (int depth = shared + 1; depth <= 5; ++depth) bytes += depth <= 4 ? 256 : (int64_t)(32 * 8 + 32 * 4);
    }
    return bytes;
}

// Utilities::calculateSizeInByteOfUnorderedMap (Utilities.hpp:1028-1040) of the taxid -> index map as
// loadContentAndFrequencyFiles fills it (Compare.hpp:127-146): the same container, so the same bucket counts.
int64_t map_bytes(const uint32_t *taxIds, uint32_t nTaxa)
{
    std::unordered_map<uint32_t, uint32_t> m;
    m.insert(std::make_pair(0u, 0u));
    for (uint32_t i = 1; i < nTaxa; ++i) m.insert(std::make_pair(taxIds[i], i));
    int64_t bytes = 0;
    for (size_t b = 0; b < m.bucket_count(); ++b) {
        const size_t s = m.bucket_size(b);
        bytes += (int64_t)sizeof(std::pair<uint32_t, uint32_t>) * (int64_t)(s ? s : 1);
    }
    return bytes;
}
} // namespace

extern "C" int kasa_refbatch_budget(const kasa_refbatch_params *p, int64_t *budget)
{
    if (!p || !budget) return KASA_E_ARG;
    try {
        const int64_t nTaxa = p->nTaxa;
        const int nK = p->kHigh - p->kLow + 1;
        const int64_t threads = p->threads > 0 ? p->threads : 1;
        int64_t avail = (p->memoryGiB > 0 ? p->memoryGiB : 5) * GIB;                  // main.cpp:590-592
        avail -= trie_bytes(p->triePrefix, p->nTrie);                                 // main.cpp:1054
        avail -= (int64_t)p->nameBytes;                                               // Compare.hpp:131
        avail -= map_bytes(p->taxIds, p->nTaxa);                                      // Compare.hpp:153
        avail -= nTaxa * 4;                                                           // Compare.hpp:154
        avail -= nTaxa * 8 * (int64_t)(p->kHigh - p->kLow);                           // Compare.hpp:155
        if (avail < 0) avail = GIB;                                                   // Compare.hpp:157-160
        const bool halvedFile = p->recordBytes == 6;
        if (p->ram) {                                                                 // Compare.hpp:183-271
            const bool half = halvedFile || (p->kLow > 6 && nTaxa <= 65535 && p->kHigh <= 12);
            const int64_t elem = half ? 6 : (p->recordBytes == 20 ? 20 : 12);
            const int64_t need = (int64_t)p->nRecords * elem;
            if (avail - need >= 0) avail -= need;                                     // else: falls back to disk mode without charging
        } else {                                                                      // Compare.hpp:277-326: four cache pages per thread
            const int64_t block = p->recordBytes == 20 ? 2048000 : 2101248;           // MetaHeader.h:137-141
            avail -= threads * block * 4 * 4;
        }
        // Compare.hpp:2803-2818
        const int64_t bitArray = (int64_t)(uint32_t)(((nTaxa + 63) >> 6) * 8 + 48 + 8 * nTaxa);   // sBitArray::sizeInBytes, BitArray.hpp:134
        const int64_t usage = (p->identifyMultiple ? GIB / threads : GIB) + threads * nK * nTaxa * 24 + threads * bitArray + 14399756 + 4 * nTaxa;
        *budget = avail > usage ? avail - usage : avail;
        return KASA_OK;
    } catch (...) {
        return KASA_E_NOMEM;
    }
}

// What one sequence (a read, or one mate of a pair) takes from the budget (Read.hpp:612-630): per strand its k-mer records
// and its padded text.  The geometry is the reader's: padding to K letters (Read.hpp:633-654), marker of (K - kLow)
// letters (Read.hpp:1068-1078), k-mer count (Read.hpp:36-57).  mode: 0 = DNA in 3 or 6 frames, 1 = --one, 2 = amino acids.
extern "C" int64_t kasa_refbatch_sequence_cost(int K, int kLow, int mode, int strands, int64_t rawLen, int coherence)
{
    const int64_t k = K, unit = mode == 2 ? 1 : 3;
    const int64_t marker = unit * (k - (int64_t)kLow);
    int64_t body = rawLen;
    if (rawLen > 0 && body + marker < unit * k) body = unit * k - marker;
    const int64_t L = body + marker;
    int64_t cnt;
    if (mode == 2) cnt = L > k + 1 ? L - k + 1 : 0;
    else if (mode == 1) { const int64_t t = L / 3; cnt = t > k + 1 ? t - k + 1 : 0; }
    else cnt = L > 3 * k + 1 ? L - 3 * k + 1 : 0;
    // sizeof(InputType::staTuple) = tuple<u64, intType, u32, u32>, or with --coherence sizeof(ppTuple) = tuple<u64, intType, u32,
    // u32, u32, u8> (MetaHeader.h:167-173,222); intType = u64 or the two-word uint128_t
    const int64_t elem = coherence ? (K > 12 ? 40 : 32) : (K > 12 ? 32 : 24);
    return (int64_t)strands * (cnt * elem + L + 16);
}

// ... and what the read as a whole takes when per-read results are kept (-q or --filter; Read.hpp:1165-1195): its entry in
// vReadNameAndLength and its row of the score matrix.  nameLen = header without its first character plus one space
// (paired-end: both mates' specifiers).
extern "C" int64_t kasa_refbatch_read_overhead(int64_t nameLen, uint32_t nTaxa, int coherence)
{
    return (int64_t)sizeof(std::pair<std::string, uint32_t>) + nameLen + 4 + (int64_t)nTaxa * 4 + (coherence ? 4 : 0);   // Read.hpp:1191-1193
}

// Number of reads of the next batch: reads are taken while more than 100 MiB of the budget are left (Read.hpp:1147).
// `firstBatch`: the budget shrinks by 0.1 % once after the first batch (Compare.hpp:3129-3132).
extern "C" uint64_t kasa_refbatch_cut(int64_t budget, int firstBatch, const int64_t *cost, uint64_t nReads)
{
    int64_t left = budget;
    if (!firstBatch && budget - (int64_t)(budget * 0.001) > 0) left -= (int64_t)(budget * 0.001);
    uint64_t n = 0;
    while (n < nReads && left > 100ll * 1024 * 1024) { left -= cost[n]; ++n; }
    return n;
}

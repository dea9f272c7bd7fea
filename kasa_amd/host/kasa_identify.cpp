// kasa_identify -- C++ host driver for `kASA identify` on top of the C ABI in include/kasa_hip.h.
//
// Keeps the reference's CLI surface for this mode (source/main.cpp:303-586, :979-1116): the same flags,
// the same index / trie / frequency / content files, the same per-read (JSON / JSONL / TSV / Kraken) and
// profile (CSV) outputs, "OUT:" / "ERROR:" prefixes and exit codes.  What Compare::CompareWithLib_partialSort
// (source/modes/Compare.hpp:2733) does per batch on the CPU is delegated to libkasa_hip.so; everything in this
// file is host logic: argument parsing, file formats, FASTA/FASTQ reading, ranking, text.
//
// Modes: identify and identify_multiple (main.cpp:979-1334); --devices a,b,... shards the batches of a file over several GPUs
// (index replicated, one RCCL all-reduce of the profile tables).  Input is streamed in chunks, batches are cut where the
// reference cuts them (-m) and parsed / computed / written in a pipeline.
// Not supported here (reported as errors, never silently ignored): --visualize; --coherence together with -e or paired-end input.  128-bit indices (build --kH 25) are read as they are (20-byte records).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <tuple>
#include <vector>

#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <rccl/rccl.h>

#include "../../include/kasa_hip.h"

using std::string;
using std::vector;

static void throwLast() { throw std::runtime_error(kasa_last_error()); }

// ---------------------------------------------------------------------------------------------------
// number text: Grisu2 as source/utils/dToStr.h prints doubles (same grid, same rounding, same prettify)
// ---------------------------------------------------------------------------------------------------
namespace numtext {
struct DiyFp { uint64_t f; int e; };
static const DiyFp kPowers[87] = {
#include "grisu_powers.inc"
};
static inline DiyFp mul(DiyFp a, DiyFp b)
{
    const unsigned __int128 p = (unsigned __int128)a.f * b.f;
    uint64_t h = (uint64_t)(p >> 64);
    if ((uint64_t)p & (1ull << 63)) ++h;
    return {h, a.e + b.e + 64};
}
static inline DiyFp normalize(uint64_t f, int e)
{
    const int s = __builtin_clzll(f);
    return {f << s, e - s};
}
static void grisuRound(char *buf, int len, uint64_t delta, uint64_t rest, uint64_t tenKappa, uint64_t wpw)
{
    while (rest < wpw && delta - rest >= tenKappa && (rest + tenKappa < wpw || wpw - rest > rest + tenKappa - wpw)) {
        buf[len - 1]--;
        rest += tenKappa;
    }
}
static int countDigits32(uint32_t n)
{
    int d = 1;
    for (uint32_t lim = 10; d < 10 && n >= lim; lim *= 10) ++d;
    return d;
}
static void grisu2(double value, char *buf, int *length, int *K)
{
    static const uint32_t kPow10[] = {1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000, 1000000000};
    uint64_t bits; memcpy(&bits, &value, 8);
    const int biased = (int)((bits >> 52) & 0x7FF);
    const uint64_t frac = bits & ((1ull << 52) - 1);
    uint64_t f; int e;
    if (biased) { f = frac | (1ull << 52); e = biased - 1075; } else { f = frac; e = -1074; }
    DiyFp pl = normalize((f << 1) + 1, e - 1);
    uint64_t mf; int me;
    if (f == (1ull << 52)) { mf = (f << 2) - 1; me = e - 2; } else { mf = (f << 1) - 1; me = e - 1; }
    DiyFp mi = {mf << (me - pl.e), pl.e};
    const double dk = (-61 - pl.e) * 0.30102999566398114 + 347;
    int k = (int)dk;
    if (k != dk) ++k;
    const unsigned index = (unsigned)((k >> 3) + 1);
    *K = -(-348 + (int)(index << 3));
    const DiyFp c = kPowers[index];
    const DiyFp W = mul(normalize(f, e), c);
    DiyFp Wp = mul(pl, c), Wm = mul(mi, c);
    Wm.f++; Wp.f--;
    uint64_t delta = Wp.f - Wm.f;
    const int oneE = Wp.e;
    const uint64_t oneF = 1ull << -oneE;
    const uint64_t wpw = Wp.f - W.f;
    uint32_t p1 = (uint32_t)(Wp.f >> -oneE);
    uint64_t p2 = Wp.f & (oneF - 1);
    int kappa = countDigits32(p1);
    int len = 0;
    while (kappa > 0) {
        const uint32_t div = kPow10[kappa - 1];
        const uint32_t d = p1 / div;
        p1 %= div;
        if (d || len) buf[len++] = (char)('0' + d);
        --kappa;
        const uint64_t tmp = ((uint64_t)p1 << -oneE) + p2;
        if (tmp <= delta) {
            *K += kappa;
            grisuRound(buf, len, delta, tmp, (uint64_t)kPow10[kappa] << -oneE, wpw);
            *length = len;
            return;
        }
    }
    for (;;) {
        p2 *= 10; delta *= 10;
        const char d = (char)(p2 >> -oneE);
        if (d || len) buf[len++] = (char)('0' + d);
        p2 &= oneF - 1;
        --kappa;
        if (p2 < delta) {
            *K += kappa;
            grisuRound(buf, len, delta, p2, oneF, wpw * (-kappa < 10 ? kPow10[-kappa] : 0));
            *length = len;
            return;
        }
    }
}
static void dtoa(double value, string &out)
{
    if (std::isnan(value)) { out += "NaN"; return; }
    if (std::isinf(value)) { out += "inf"; return; }
    if (value == 0) { out += "0.0"; return; }
    if (value < 0) { out += '-'; value = -value; }
    char d[32]; int n, k;
    grisu2(value, d, &n, &k);
    const int kk = n + k;
    auto expo = [&](int x) { if (x < 0) { out += '-'; x = -x; } out += std::to_string(x); };
    if (n <= kk && kk <= 21) { out.append(d, n); out.append((size_t)(kk - n), '0'); out += ".0"; }
    else if (0 < kk && kk <= 21) { out.append(d, kk); out += '.'; out.append(d + kk, n - kk); }
    else if (-6 < kk && kk <= 0) { out += "0."; out.append((size_t)(-kk), '0'); out.append(d, n); }
    else if (n == 1) { out += d[0]; out += 'e'; expo(kk - 1); }
    else { out += d[0]; out += '.'; out.append(d + 1, n - 1); out += 'e'; expo(kk - 1); }
}
static void itoa(uint64_t v, string &out) { out += std::to_string(v); }
} // namespace numtext

// ---------------------------------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------------------------------
struct Content { vector<string> names; vector<uint32_t> taxids; };

static vector<string> splitTabs(const string &s)
{
    vector<string> out; size_t a = 0;
    for (;;) { const size_t b = s.find('\t', a); out.push_back(s.substr(a, b == string::npos ? b : b - a)); if (b == string::npos) break; a = b + 1; }
    return out;
}

static Content loadContent(const string &path) // Compare.hpp:111-151
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("The content file or the frequency file cannot be found!");
    Content c; c.names.push_back("non_unique"); c.taxids.push_back(0);
    string line; bool asStr = false;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        const auto cols = splitTabs(line);
        if (cols.size() >= 5) asStr = true;
        if (cols.size() < 4) throw std::runtime_error("Content file contains less than 4 columns, it may be damaged... The faulty line was: " + line + "\n");
        string nm = cols[0]; nm.erase(std::remove(nm.begin(), nm.end(), ','), nm.end());
        c.names.push_back(nm);
        c.taxids.push_back((uint32_t)std::stoul(asStr ? cols[4] : cols[1]));
    }
    return c;
}

// Compare.hpp:166-179: freq[l * nTaxa + t] = k-mers of taxon t at k = kHigh - l
static vector<uint64_t> loadFreq(const string &prefix, size_t nTaxa, int kHigh, int kLow)
{
    std::ifstream f(prefix + "_f.txt");
    if (!f) throw std::runtime_error("The content file or the frequency file cannot be found!");
    const int nK = kHigh - kLow + 1;
    vector<uint64_t> out(nTaxa * (size_t)nK, 0);
    string line; size_t row = 0;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        const auto cols = splitTabs(line);
        const size_t numK = cols.size() - 1;
        for (int l = 0; l < nK && row < nTaxa; ++l) {
            const size_t col = 1 + numK - (size_t)(kHigh - l);
            if (col >= 1 && col < cols.size()) out[(size_t)l * nTaxa + row] = std::stoull(cols[col]);
        }
        ++row;
    }
    return out;
}

// KASA_HOST_TIMING=1: where the host spends the time of "Time fastq" (seconds, summed over the file)
struct HostTimers { double read = 0, cut = 0, parse = 0, merge = 0, form = 0, write = 0, upload = 0, compute = 0, rank = 0, text = 0, fetch = 0, encode = 0, sort = 0, score = 0; std::mutex mu; bool on = getenv("KASA_HOST_TIMING") != nullptr; };
static HostTimers g_ht;
static std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
static void mark(const char *what, uint64_t id = ~0ull)          // KASA_HOST_TIMING: a time line of the file's pipeline
{
    if (!g_ht.on) return;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count();
    std::lock_guard<std::mutex> lk(g_ht.mu);
    if (id == ~0ull) fprintf(stderr, "kasa: t=%.3f %s\n", t, what); else fprintf(stderr, "kasa: t=%.3f %s %llu\n", t, what, (unsigned long long)id);
}
struct ScopedTimerMt {                               // from several threads: summed under a lock (CPU seconds, not wall time)
    double &acc; std::mutex &mu; std::chrono::steady_clock::time_point t0;
    ScopedTimerMt(double &a, std::mutex &m) : acc(a), mu(m), t0(std::chrono::steady_clock::now()) {}
    ~ScopedTimerMt() { const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); std::lock_guard<std::mutex> lk(mu); acc += d; }
};
struct ScopedTimer {
    double &acc; std::chrono::steady_clock::time_point t0;
    explicit ScopedTimer(double &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopedTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// Large host arrays.  Fresh 4 KB pages are touched at 6 GB/s by one core and 17 GB/s by sixteen on the GPU box, huge pages at
// 17 and 195 GB/s (tools/hostio_probe.cpp), and a std::vector zeroes what it resizes on ONE core: arrays that hold a batch
// are anonymous mappings with huge pages asked for, never value-initialised, grown in place (mremap), and filled by the
// threads that parse.
template <class T> struct HugeVec {
    T *p = nullptr; size_t n = 0, cap = 0;
    HugeVec() {}
    ~HugeVec() { release(); }
    HugeVec(const HugeVec &) = delete;
    HugeVec &operator=(const HugeVec &) = delete;
    HugeVec(HugeVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    HugeVec &operator=(HugeVec &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
    static size_t mapBytes(size_t elems) { const size_t g = (size_t)2 << 20; return (elems * sizeof(T) + g - 1) / g * g; }
    void release() { if (p) ::munmap((void *)p, mapBytes(cap)); p = nullptr; n = cap = 0; }
    void reserve(size_t want)
    {
        if (want <= cap) return;
        const size_t nb = mapBytes(std::max(want, cap + cap / 2));
        void *q = p ? ::mremap((void *)p, mapBytes(cap), nb, MREMAP_MAYMOVE) : ::mmap(nullptr, nb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        (void)::madvise(q, nb, MADV_HUGEPAGE);
        p = (T *)q; cap = nb / sizeof(T);
    }
    void resize(size_t k) { reserve(k); n = k; }                // (new elements are NOT initialised)
    void clear() { n = 0; }
    void push_back(const T &v) { if (n == cap) reserve(n + 1); p[n++] = v; }
    void append(const T *src, size_t k) { if (n + k > cap) reserve(n + k); std::memcpy((void *)(p + n), (const void *)src, k * sizeof(T)); n += k; }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T &back() { return p[n - 1]; }
};

// memcpy by several threads (a batch's bases are gigabytes)
static void parCopy(void *dst, const void *src, size_t bytes, unsigned threads)
{
    const size_t piece = (size_t)32 << 20;
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads ? threads : 1, bytes / piece));
    if (nt <= 1) { if (bytes) std::memcpy(dst, src, bytes); return; }
    vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([=] { const size_t a = bytes / nt * t, e = t + 1 == nt ? bytes : bytes / nt * (t + 1); std::memcpy((char *)dst + a, (const char *)src + a, e - a); });
    for (auto &th : pool) th.join();
}

// The reads of a chunk / batch: bases and specifiers back to back with running offsets (no object per read)
struct ReadSet {
    HugeVec<uint8_t> bases; HugeVec<int64_t> off;              // off[nSequences + 1], off[0] = 0
    HugeVec<char> nameBlob; HugeVec<uint64_t> nameOff;         // specifier of read r = nameBlob[nameOff[r] .. nameOff[r + 1])
    HugeVec<uint32_t> lengths;                                 // "Length" of read r
    bool protein = false, fasta = true;
    // A record of a million letters and more: how the reference's reader cuts its lines (Utilities::FileReader hands out text
    // up to the next line feed or the end of its 2048-byte buffer, Utilities.hpp:448-539) -- it reads such a record in
    // pieces that end with one of those calls (Read.hpp:371-600; Batcher::piecesOf).
    struct LongRec { size_t read; vector<uint32_t> letters; vector<uint8_t> feed; };
    vector<LongRec> longRecs;
    ReadSet() { off.push_back(0); nameOff.push_back(0); }
    ReadSet(ReadSet &&) = default;
    ReadSet &operator=(ReadSet &&) = default;
    size_t size() const { return lengths.size(); }
    size_t nameLen(size_t r) const { return (size_t)(nameOff[r + 1] - nameOff[r]); }
    std::string_view name(size_t r) const { return std::string_view(nameBlob.data() + nameOff[r], nameLen(r)); }
    void clear() { bases.clear(); off.clear(); off.push_back(0); nameBlob.clear(); nameOff.clear(); nameOff.push_back(0); lengths.clear(); longRecs.clear(); }
    // reads [first, last) as a set of their own (spr sequences per read; offsets rebased); `threads` copy the bases
    ReadSet slice(size_t first, size_t last, size_t spr, unsigned threads) const
    {
        ReadSet o;
        o.protein = protein; o.fasta = fasta;
        for (const LongRec &lr : longRecs) if (lr.read >= first && lr.read < last) { o.longRecs.push_back(lr); o.longRecs.back().read -= first; }
        const size_t m = last - first;
        const int64_t s0 = off[first * spr], s1 = off[last * spr];
        o.bases.resize((size_t)(s1 - s0)); parCopy(o.bases.data(), bases.data() + s0, (size_t)(s1 - s0), threads);
        o.off.resize(m * spr + 1);
        for (size_t q = 0; q <= m * spr; ++q) o.off[q] = off[first * spr + q] - s0;
        const uint64_t n0 = nameOff[first], n1 = nameOff[last];
        o.nameBlob.resize((size_t)(n1 - n0)); if (n1 > n0) std::memcpy(o.nameBlob.data(), nameBlob.data() + n0, (size_t)(n1 - n0));
        o.nameOff.resize(m + 1);
        for (size_t r = 0; r <= m; ++r) o.nameOff[r] = nameOff[first + r] - n0;
        o.lengths.resize(m); if (m) std::memcpy(o.lengths.data(), lengths.data() + first, m * 4);
        return o;
    }
    // the reads of `q` behind mine (offsets rebased); `threads` copy the bases
    void appendSet(const ReadSet &q, unsigned threads)
    {
        const size_t b0 = bases.size(), n0 = nameBlob.size(), r0 = size(), s0 = off.size() - 1, ns = q.off.size() - 1;
        for (const LongRec &lr : q.longRecs) { longRecs.push_back(lr); longRecs.back().read += r0; }
        bases.resize(b0 + q.bases.size()); parCopy(bases.data() + b0, q.bases.data(), q.bases.size(), threads);
        nameBlob.append(q.nameBlob.data(), q.nameBlob.size());
        off.resize(s0 + ns + 1); nameOff.resize(r0 + q.size() + 1); lengths.resize(r0 + q.size());
        for (size_t i = 0; i < ns; ++i) off[s0 + 1 + i] = (int64_t)b0 + q.off[i + 1];
        for (size_t r = 0; r < q.size(); ++r) { nameOff[r0 + 1 + r] = n0 + q.nameOff[r + 1]; lengths[r0 + r] = q.lengths[r]; }
    }
};

// kASA::detectAlphabet (kASA.hpp:155-183) on the first four characters of the file's second line
// (Utilities::getFirstSequenceOfFile, Utilities.hpp:137-143)
static bool detectProtein(const char *data, size_t size, bool verbose)
{
    const char *nl = (const char *)memchr(data, '\n', size);
    string four;
    if (nl) {
        const size_t a = (size_t)(nl - data);
        const char *nl2 = (const char *)memchr(data + a + 1, '\n', size - a - 1);
        const size_t b = nl2 ? (size_t)(nl2 - data) : size;
        four.assign(data + a + 1, std::min<size_t>(4, b - a - 1));
    }
    auto in = [](const string &s, const char *set) { if (s.empty()) return false; for (char c : s) if (!strchr(set, toupper((unsigned char)c)) || c == 0) return false; return true; };
    if (in(four, "ACGTURYKMSWBDHVN-")) { if (verbose) std::cout << "OUT: DNA sequences detected." << std::endl; return false; }
    if (!in(four, "ABCDEFGHIJKLMNOPQRSTUVWXYZ*-"))
        std::cerr << "ERROR: The sequence is neither recognized as protein nor DNA. It will be treated as protein sequence but it may fail... Sequence was: " << four << std::endl;
    else if (verbose) std::cout << "OUT: Protein sequences detected." << std::endl;
    return true;
}

// One run of whole records, data[begin, end) with begin at a header line: what Read.hpp:699-760 hands on, for reads that
// fit one chunk.  A '\r' stays part of its line, as with the reference's getline.
// shorter records are one piece whatever the options (100 MiB / 48 B / 2 strands); tests lower both numbers
static const size_t kLongSequence = getenv("KASA_LONG_SEQUENCE") ? (size_t)atoll(getenv("KASA_LONG_SEQUENCE")) : 1000000;
static const int64_t kPieceBytes = getenv("KASA_PIECE_BYTES") ? (int64_t)atoll(getenv("KASA_PIECE_BYTES")) : 100ll * 1024 * 1024;   // Read.hpp:438,507

static void parseRecords(const char *data, size_t begin, size_t end, bool fasta, ReadSet &rs, size_t streamBase)
{
    rs.fasta = fasta;
    size_t a = begin;
    auto nextLine = [&](size_t &lb, size_t &le) -> bool {          // [lb, le) without the line feed
        if (a >= end) return false;
        lb = a;
        const void *nl = memchr(data + a, '\n', end - a);
        le = nl ? (size_t)((const char *)nl - data) : end;
        a = le + 1;
        return true;
    };
    // room for what the run can hold at most (address space: untouched pages cost nothing)
    rs.bases.reserve(rs.bases.size() + (end - begin)); rs.nameBlob.reserve(rs.nameBlob.size() + (end - begin) + 16);
    size_t lb, le;
    bool have = nextLine(lb, le);
    while (have) {
        if (lb == le) { have = nextLine(lb, le); continue; }
        rs.nameBlob.append(data + lb + 1, le - lb - 1);             // Read.hpp:711-714: header without its first character
        rs.nameBlob.push_back(' ');                                 // ... plus a trailing space
        rs.nameOff.push_back(rs.nameBlob.size());
        uint32_t nLines = 0; const size_t b0 = rs.bases.size(), seqBegin = a;
        while ((have = nextLine(lb, le))) {
            const bool empty = lb == le;
            if (!empty && data[lb] == (fasta ? '>' : '+')) break;
            if (!empty) rs.bases.append((const uint8_t *)data + lb, le - lb);
            if (!empty || !fasta) ++nLines;
        }
        const size_t len = rs.bases.size() - b0, seqEnd = have ? lb : end;
        if (len >= kLongSequence) {
            // the getChunk calls over the record's lines: each ends at a line feed or at the next multiple of 2048 bytes of the file
            ReadSet::LongRec lr; lr.read = rs.lengths.size();
            for (size_t pos = seqBegin; pos < seqEnd;) {
                const size_t bufEnd = ((streamBase + pos) / 2048 + 1) * 2048 - streamBase, limit = std::min(bufEnd, seqEnd);
                const void *nl = memchr(data + pos, '\n', limit - pos);
                if (nl) { const size_t q = (size_t)((const char *)nl - data); lr.letters.push_back((uint32_t)(q - pos)); lr.feed.push_back(1); pos = q + 1; }
                else if (limit == bufEnd) { lr.letters.push_back((uint32_t)(limit - pos)); lr.feed.push_back(0); pos = limit; }
                else { lr.letters.push_back((uint32_t)(seqEnd - pos)); lr.feed.push_back(1); pos = seqEnd; }   // the file ends without a line feed: the reader supplies one
            }
            rs.longRecs.push_back(std::move(lr));
        }
        for (size_t k = b0; k < rs.bases.size(); ++k)
            if (rs.bases[k] == ' ' || rs.bases[k] == '\t') throw std::runtime_error("Spaces or tabs inside read, please check your input."); // Read.hpp:659
        if (!fasta) {                                               // the '+' line is current: quality lines follow
            size_t q = 0;
            while ((have = nextLine(lb, le)) && q < len) q += le - lb;
            if (q > len) throw std::runtime_error("Quality string and DNA string do not have the same length!");
        }
        rs.off.push_back((int64_t)rs.bases.size());
        rs.lengths.push_back((uint32_t)(len + nLines));             // one extra per sequence line (Read.hpp:723-731)
    }
}

// First record start at or after `from` that is safe to cut at (N1: the input is parsed by several threads).  FASTA: a
// line starting with '>'.  FASTQ: a line starting with '@' whose third line starts with '+' and whose fourth line is as
// long as its second -- a quality line that happens to start with '@' fails that test.  npos: none found nearby.
static size_t findRecordStart(const char *data, size_t size, size_t from, bool fasta, size_t span = 1u << 20)
{
    const size_t npos = string::npos;
    const size_t limit = span >= size - std::min(size, from) ? size : from + span;
    size_t p = from;
    while (p < limit) {
        const void *nl = memchr(data + p, '\n', limit - p);
        if (!nl) return npos;
        p = (size_t)((const char *)nl - data) + 1;
        if (p >= size) return npos;
        if (fasta) { if (data[p] == '>') return p; continue; }
        if (data[p] != '@') continue;
        size_t l[5]; l[0] = p; bool ok = true;
        for (int i = 1; i < 5 && ok; ++i) {
            const void *e = memchr(data + l[i - 1], '\n', size - l[i - 1]);
            if (!e) { ok = (i == 4); l[i] = size + 1; break; }
            l[i] = (size_t)((const char *)e - data) + 1;
        }
        if (!ok || l[2] >= size || data[l[2]] != '+') continue;
        if (l[4] - l[3] != l[2] - l[1]) continue;                  // quality as long as the sequence (both with their '\n')
        if (l[4] < size && data[l[4]] != '@') continue;
        return p;
    }
    return npos;
}

// A piece of whole records parsed by several threads and put behind the reads `out` holds: the piece is cut into one run per
// thread at safe record starts, every run is parsed into a set of its own (`parts`: kept by the caller over its pieces, so
// that their pages are touched once), and the same threads copy the runs to their places in `out`.
static void parsePiece(const char *data, size_t size, bool fasta, unsigned threads, size_t minRun, ReadSet &out, vector<ReadSet> &parts, size_t streamBase)
{
    if (const char *e = getenv("KASA_PARSE_CHUNK")) minRun = std::max<size_t>(1, (size_t)atoll(e));   // tests force small runs
    const size_t want = std::max<size_t>(1, std::min<size_t>(threads ? threads : 1, size / minRun));
    vector<size_t> cut{0};
    for (size_t c = 1; c < want; ++c) {
        const size_t p = findRecordStart(data, size, std::max(cut.back(), size / want * c), fasta);
        if (p != string::npos && p > cut.back()) cut.push_back(p);
    }
    cut.push_back(size);
    const size_t nc = cut.size() - 1;
    auto inParallel = [&](const std::function<void(size_t)> &fn) {
        if (nc == 1) { fn(0); return; }
        vector<std::exception_ptr> err(nc);
        vector<std::thread> pool;
        for (size_t c = 0; c < nc; ++c) pool.emplace_back([&, c] { try { fn(c); } catch (...) { err[c] = std::current_exception(); } });
        for (auto &t : pool) t.join();
        for (auto &e : err) if (e) std::rethrow_exception(e);          // like Compare.hpp:3312-3314
    };
    if (nc == 1 && out.size() == 0 && out.bases.empty()) { ScopedTimer tm(g_ht.parse); parseRecords(data, 0, size, fasta, out, streamBase); return; }
    if (parts.size() < nc) parts.resize(nc);
    { ScopedTimer tm(g_ht.parse); inParallel([&](size_t c) { parts[c].clear(); parseRecords(data, cut[c], cut[c + 1], fasta, parts[c], streamBase); }); }
    ScopedTimer tmMerge(g_ht.merge);
    // the runs' reads behind the pending ones: places from running sums, copies by the same threads
    vector<size_t> b0(nc), r0(nc), m0(nc);
    size_t nb = out.bases.size(), nr = out.size(), nm = out.nameBlob.size();
    for (size_t c = 0; c < nc; ++c) { b0[c] = nb; r0[c] = nr; m0[c] = nm; nb += parts[c].bases.size(); nr += parts[c].size(); nm += parts[c].nameBlob.size(); }
    out.bases.resize(nb); out.off.resize(nr + 1); out.nameBlob.resize(nm); out.nameOff.resize(nr + 1); out.lengths.resize(nr);
    out.fasta = fasta;
    for (size_t c = 0; c < nc; ++c) for (ReadSet::LongRec &lr : parts[c].longRecs) { out.longRecs.push_back(std::move(lr)); out.longRecs.back().read += r0[c]; }
    inParallel([&](size_t c) {
        const ReadSet &q = parts[c];
        if (!q.bases.empty()) memcpy(out.bases.data() + b0[c], q.bases.data(), q.bases.size());
        if (!q.nameBlob.empty()) memcpy(out.nameBlob.data() + m0[c], q.nameBlob.data(), q.nameBlob.size());
        for (size_t r = 0; r < q.size(); ++r) {
            out.lengths[r0[c] + r] = q.lengths[r];
            out.off[r0[c] + r + 1] = (int64_t)b0[c] + q.off[r + 1];
            out.nameOff[r0[c] + r + 1] = m0[c] + q.nameOff[r + 1];
        }
    });
}

static ReadSet readInput(const string &path, bool verbose, unsigned threads)
{
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) throw std::runtime_error("Input file not found");
    string data;
    { struct stat st; if (stat(path.c_str(), &st) == 0 && st.st_size > 0) data.reserve((size_t)st.st_size + 16); }
    vector<char> buf(1 << 22); int n;
    while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) data.append(buf.data(), (size_t)n);
    gzclose(g);
    ReadSet rs;
    if (data.empty()) return rs;
    if (data[0] != '>' && data[0] != '@') throw std::runtime_error("Input does not start with @ or >.");
    const bool fasta = data[0] == '>';
    rs.protein = detectProtein(data.data(), data.size(), verbose);
    vector<ReadSet> parts;
    parsePiece(data.data(), data.size(), fasta, threads, 8u << 20, rs, parts, 0);
    return rs;
}

// ---------------------------------------------------------------------------------------------------
// ranking + text (Compare.hpp:1452-1890) and profile (Compare.hpp:3466-3665)
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// input: streamed in chunks of whole records (N1) -- never the whole file in memory
// ---------------------------------------------------------------------------------------------------
// Reads a FASTA/FASTQ(.gz) file block by block (gz members inflate as they come) and hands out chunks that end at a record
// boundary; a chunk is parsed by all host threads (parseRecords over runs cut at safe record starts).
struct ChunkReader {
    gzFile g = nullptr;
    int fd = -1;                                  // plain (not gzip'ed) input is read with pread(2): zlib's pass-through copies at 1 GB/s
    // Two buffers that take turns: the block read behind what the last chunk left over; the chunk handed out stays where it is
    // while its tail moves to the other buffer.  Their pages are touched once (8 threads pread into a warm buffer at 70 GB/s;
    // into fresh memory the same call is bound by the page faults).
    HugeVec<char> buf[2]; int cur = 0; size_t have = 0;
    bool fasta = false, protein = false, eof = false, first = true;
    size_t blockBytes = 256u << 20;
    explicit ChunkReader(const string &path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("Input file not found");
        unsigned char magic[2] = {0, 0};
        const ssize_t m = ::pread(fd, magic, 2, 0);
        if (m == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            g = gzdopen(fd, "rb");                 // takes the descriptor over
            if (!g) { ::close(fd); fd = -1; throw std::runtime_error("Input file not found"); }
            fd = -1;
            gzbuffer(g, 1u << 20);
        }
        if (const char *e = getenv("KASA_READ_BLOCK")) blockBytes = std::max<size_t>(1, (size_t)atoll(e));   // tests force small blocks
    }
    ~ChunkReader() { if (g) gzclose(g); if (fd >= 0) ::close(fd); }
    off_t filePos = 0, fileSize = -1;
    size_t handedOut = 0, chunkStart = 0;         // where the chunk last handed out starts in the (inflated) file
    long readSome(char *dst, size_t want)
    {
        if (g) return gzread(g, dst, (unsigned)std::min<size_t>(want, 1u << 30));
        // a plain file: its bytes are copied out of the page cache by several threads at known offsets
        if (fileSize < 0) { struct stat st; fileSize = (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) ? st.st_size : 0; }
        if (fileSize == 0) return (long)::read(fd, dst, std::min<size_t>(want, 1u << 30));          // a pipe or the like
        want = (size_t)std::min<off_t>((off_t)std::min<size_t>(want, 1u << 30), fileSize - filePos);
        if (want == 0) return 0;
        const size_t nt = std::max<size_t>(1, std::min<size_t>(8, want >> 24));
        const size_t slice = (want + nt - 1) / nt;
        vector<long> got(nt, 0);
        auto one = [&](size_t t) {
            size_t a = t * slice, e = std::min(want, a + slice);
            while (a < e) { const ssize_t n = ::pread(fd, dst + a, e - a, filePos + (off_t)a); if (n <= 0) break; a += (size_t)n; got[t] += n; }
        };
        if (nt == 1) one(0);
        else { vector<std::thread> pool; for (size_t t = 0; t < nt; ++t) pool.emplace_back(one, t); for (auto &th : pool) th.join(); }
        long total = 0;
        for (size_t t = 0; t < nt; ++t) { total += got[t]; if ((size_t)got[t] < std::min(want, (t + 1) * slice) - t * slice) break; }   // (a short slice ends the data)
        filePos += total;
        return total;
    }
    // next chunk of whole records (false at the end of the file); it stays valid until the call after the next one
    bool next(const char *&chunk, size_t &size, bool verbose)
    {
        chunk = nullptr; size = 0;
        while (!eof) {
            HugeVec<char> &D = buf[cur];
            size_t got = 0;
            {
                ScopedTimer tm(g_ht.read);
                // (the first block is a quarter: the reads in it size the device buffers, which are then allocated while the
                // rest of the first batch is parsed)
                const size_t want = first ? std::max<size_t>(1, blockBytes / 4) : blockBytes;
                D.reserve(have + blockBytes);
                while (got < want) {
                    const long n = readSome(D.data() + have + got, want - got);
                    if (n <= 0) { eof = true; break; }
                    got += (size_t)n;
                }
                have += got;
            }
            ScopedTimer tmCut(g_ht.cut);
            if (first && have > 0 && !eof) {                          // the alphabet is read off the second line's first four characters: wait for them
                const char *nl = (const char *)memchr(D.data(), '\n', have);
                if (!nl || have - (size_t)(nl - D.data()) - 1 < 4) continue;
            }
            if (first && have > 0) {
                if (D[0] != '>' && D[0] != '@') throw std::runtime_error("Input does not start with @ or >.");
                fasta = D[0] == '>';
                protein = detectProtein(D.data(), have, verbose);
                first = false;
            }
            if (eof) break;
            // cut at the last safe record start; what follows waits for the next block
            // (looked for in the last 4 MiB first; a record longer than that -- a contig, a long read -- sends the search over
            // the whole buffer instead of letting block after block pile up behind it)
            size_t cut = string::npos, from = have > (4u << 20) ? have - (4u << 20) : 0;
            for (int pass = 0; pass < 2 && cut == string::npos; ++pass) {
                for (;;) {
                    const size_t p = findRecordStart(D.data(), have, from, fasta, have - from);
                    if (p == string::npos) break;
                    cut = p; from = p;
                }
                if (from == 0) break;
                if (cut == string::npos) from = 0;
            }
            if (cut == string::npos || cut == 0) continue;          // ONE record so far: keep reading behind what is there
            HugeVec<char> &T = buf[cur ^ 1];
            T.reserve(have - cut + blockBytes);
            std::memcpy(T.data(), D.data() + cut, have - cut);
            chunk = D.data(); size = cut;
            have -= cut; cur ^= 1;
            chunkStart = handedOut; handedOut += size;
            return true;
        }
        if (have > 0) { chunk = buf[cur].data(); size = have; have = 0; chunkStart = handedOut; handedOut += size; return true; }
        return false;
    }
};


// ---------------------------------------------------------------------------------------------------
// one input file (CompareWithLib_partialSort, Compare.hpp:2733-3766) over one or several devices
// ---------------------------------------------------------------------------------------------------
struct Params {
    string mode = "identify";
    string content, index, input, input2, rtt, profile;   // input2: second file of paired-end input (-1 / -2)
    int kHigh = 12, kLow = 7, beasts = 3, frames = 3, K = 12;   // K: letters per index k-mer (25 for a 128-bit index)
    vector<int> devices{0};                      // --device d / --devices a,b,...: read shards go to the devices in turn (index replicated)
    bool kSetByUser = false;
    string codonFile, codonId;                   // -a/--alphabet <gc.prt> <id>
    bool filter = false; string filterClean, filterCont; float errorThreshold = 0.5f;   // --filter <clean> <contaminants>, --errorThreshold
    unsigned threads = 0;                       // -n: host threads for parsing and text output (0: all cores, at most 32)
    bool hostRank = false;                         // --host-rank: the whole CSR comes back and the host ranks every read
    bool hostText = false;                         // --host-text: the hits come back and the host writes the per-read text
    bool allowDeviceSplit = false;                 // --allow-device-split: a reference batch larger than the device holds is cut (last digits of scores may differ)
    int memoryGiB = 0, refThreads = 1; bool ram = false;   // -m / -n / -r as the reference's batch budget sees them (kasa_refbatch_*)
    float threshold = 0.f;
    enum Fmt { Kraken, Json, JsonL, Tsv } fmt = Json;
    bool verbose = false, coverage = false, unique = false, protein = false;
    bool gzipOut = false;                        // --gzip: the --filter files are written through zlib (Compare.hpp:2455,3713-3731)
    bool coherence = false; float coherenceThreshold = 11.0f;   // --coherence, --coherenceThreshold (MetaHeader.h:159)
};

struct IndexFiles {                               // what Compare::ReadIndex loads (Compare.hpp:49-363), shared by every worker
    Content content;
    vector<uint64_t> freqAll, freq;
    uint64_t nRec = 0; int recBytes = 12;
    vector<uint32_t> tp; vector<uint64_t> tc;
    vector<uint8_t> lut;
    uint64_t nameBytes = 0;
    vector<kasa_index *> onDevice;                // one immutable index object per device, shared by all contexts there
    // An index of 2^32 records and more (positions in an index object are 32-bit) lives on the device as several range
    // partitions, every one an index object of its own, cut between entries of the `_trie` file (a prefix range never
    // straddles a cut): parts[d][j] on device d, cuts[j] = first 30-bit prefix of partition j (cuts[0] = 0).  The reference
    // streams an index of any size from disk (Compare.hpp:286-318).  onDevice[d] = parts[d][0] then.
    vector<vector<kasa_index *>> parts;
    vector<uint64_t> cuts;
    ~IndexFiles()
    {
        if (parts.empty()) { for (auto *ix : onDevice) kasa_index_destroy(ix); }
        else for (auto &v : parts) for (auto *ix : v) kasa_index_destroy(ix);
    }
};

static float weightOf(int k) { return (float)(k * k) / 625.f; }
static float bestScore(uint64_t len, const Params &p) // Compare.hpp:1452-1481
{
    float best = 0.f;
    for (int i = p.kLow; i <= p.kHigh; ++i) {
        if (p.protein) best += (float)(len - (uint64_t)i + 1) * weightOf(i);
        else if (p.frames == 1) best += (float)(len / 3 - (uint64_t)i + 1) * weightOf(i);
        else if (p.frames == 6) best += (float)(2 * (len - (uint64_t)(i * 3) + 1)) * weightOf(i);
        else best += (float)(len - (uint64_t)(i * 3) + 1) * weightOf(i);
    }
    return best;
}

struct Writer {
    const Params &p; const Content &c; const vector<uint64_t> &freq;
    vector<std::tuple<size_t, float, double>> res;
    Writer(const Params &pp, const Content &cc, const vector<uint64_t> &ff) : p(pp), c(cc), freq(ff), res(cc.names.size()) {}

    float coherence = 0.f;           // --coherence: the score of the read being written (Compare.hpp:1662-1665,1711-1715,1792-1796)
    void obj(string &o, const std::tuple<size_t, float, double> &h, float best, bool pretty) const
    {
        using numtext::dtoa; using numtext::itoa;
        if (pretty) {
            o += "\t\t\"tax ID\": \""; itoa(c.taxids[std::get<0>(h)], o); o += "\",\n\t\t\"Name\": \""; o += c.names[std::get<0>(h)];
            o += "\",\n\t\t\"k-mer Score\": "; dtoa(std::get<1>(h), o); o += ",\n\t\t\"Relative Score\": "; dtoa(std::get<2>(h), o);
            o += ",\n\t\t\"Error\": "; dtoa((best - std::get<1>(h)) / best, o);
            if (p.coherence) { o += ",\n\t\t\"Coherence\": "; dtoa(coherence, o); }
            o += "\n\t}";
        } else {
            o += " \"tax ID\": \""; itoa(c.taxids[std::get<0>(h)], o); o += "\", \"Name\": \""; o += c.names[std::get<0>(h)];
            o += "\", \"k-mer Score\": "; dtoa(std::get<1>(h), o); o += ", \"Relative Score\": "; dtoa(std::get<2>(h), o);
            o += ", \"Error\": "; dtoa((best - std::get<1>(h)) / best, o);
            if (p.coherence) { o += ",\"Coherence\": "; dtoa(coherence, o); }
            o += "}";
        }
    }

    bool lastContaminated = false;   // --filter: the read just written comes within --errorThreshold of the perfect score
    // host ranking of one read from its full row (Compare.hpp:1501-1524)
    void read(string &o, uint64_t number, std::string_view name, uint32_t len, const uint32_t *tax, const float *score, uint64_t n)
    {
        int64_t cnt = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (!(score[i] > 0.f)) continue;
            const double rel = score[i] / (1.0 + log2(freq[tax[i]] * double(uint32_t(len - (p.protein ? p.K : p.K * 3) + 1)))); // Compare.hpp:1506-1511
            if (rel >= p.threshold) { res[cnt] = std::make_tuple((size_t)tax[i], score[i], rel); ++cnt; }
        }
        if (cnt > 0)
            std::sort(res.begin(), res.begin() + cnt, [](const std::tuple<size_t, float, double> &a, const std::tuple<size_t, float, double> &b) { return std::get<2>(a) > std::get<2>(b); });
        float maxV = 0.f;
        for (int64_t i = 0; i < cnt; ++i) maxV = std::max(maxV, std::get<1>(res[i]));
        print(o, number, name, len, cnt, maxV);
    }
    // a read ranked on the device (kasa_batch_rank): its printable hits in order, and the largest k-mer score of all its hits
    struct DeviceHit { uint32_t tax; float score; double rel; };
    void readRanked(string &o, uint64_t number, std::string_view name, uint32_t len, const DeviceHit *hits, uint32_t n, float maxV)
    {
        if (res.size() < n) res.resize(n);
        for (uint32_t i = 0; i < n; ++i) res[i] = std::make_tuple((size_t)hits[i].tax, hits[i].score, hits[i].rel);
        print(o, number, name, len, (int64_t)n, maxV);
    }
    // res[0 .. cnt) = the hits in printing order (Compare.hpp:1526-1872)
    void print(string &o, uint64_t number, std::string_view name, uint32_t len, int64_t cnt, float maxV)
    {
        lastContaminated = false;
        using numtext::dtoa; using numtext::itoa;
        const float best = bestScore(len, p);
        if (cnt == 0) {
            switch (p.fmt) {
            case Params::Tsv: itoa(number, o); o += "\t"; o += name; o += p.coherence ? "\t-\t-\t-\t-\t-\n" : "\t-\t-\t-\t-\n"; break;
            case Params::Json:
                o += number == 0 ? "{\n" : ",\n{\n"; o += "\t\"Read number\": "; itoa(number, o);
                o += ",\n\t\"Specifier from input file\": \""; o += name; o += "\",\n\t\"Length\": "; itoa(len, o);
                o += ",\n\t\"Top hits\": [\n\t],\n\t\"Further hits\": [\n\t]\n}"; break;
            case Params::JsonL:
                o += "{ \"Read number\": "; itoa(number, o); o += ", \"Specifier from input file\": \""; o += name;
                o += "\", \"Length\": "; itoa(len, o); o += ", \"Top hits\": [], \"Further hits\": [] }\n"; break;
            case Params::Kraken: o += "U\t"; o += name; o += "\t0\t"; o += (char)len; o += "\tA:00\n"; break; // one raw byte (Compare.hpp:1568)
            }
            return;
        }
        int64_t top = 1;
        for (int64_t i = 1; i < cnt && i < p.beasts; ++i) { if (std::get<1>(res[i]) / maxV > 0.8f) ++top; else break; }
        lastContaminated = (best - double(maxV)) / best < p.errorThreshold ||          // Compare.hpp:1597-1606
                           (p.coherence && coherence >= p.coherenceThreshold);
        float before = 0;
        switch (p.fmt) {
        case Params::Tsv: {
            string s1, s2, s3, s4; itoa(number, s1); s1 += "\t"; s1 += name; s1 += "\t";
            for (int64_t j = 0, i = 0; i < cnt && j < p.beasts; ++i) {
                const auto &h = res[i];
                itoa(c.taxids[std::get<0>(h)], s1); s1 += ";"; s2 += c.names[std::get<0>(h)]; s2 += ";";
                dtoa(std::get<2>(h), s3); s3 += ","; dtoa(std::get<1>(h), s3); s3 += ";";
                dtoa((best - std::get<1>(h)) / best, s4); s4 += ";";
                if (before != std::get<1>(h)) { before = std::get<1>(h); ++j; }
            }
            for (string *s : {&s1, &s2, &s3, &s4}) if (!s->empty() && s->back() == ';') s->pop_back();
            if (!s2.empty()) {
                o += s1; o += "\t"; o += s2; o += "\t"; o += s3; o += "\t"; o += s4;
                if (p.coherence) { o += "\t"; dtoa(coherence, o); }
                o += "\n";
            }
        } break;
        case Params::Json: {
            o += number == 0 ? "{\n" : ",\n{\n"; o += "\t\"Read number\": "; itoa(number, o);
            o += ",\n\t\"Specifier from input file\": \""; o += name; o += "\",\n\t\"Length\": "; itoa(len, o); o += ",\n\t\"Top hits\": [\n";
            for (int64_t i = 0; i < top; ++i) { o += i == 0 ? "\t{\n" : ",\n\t{\n"; obj(o, res[i], best, true); }
            o += "\n\t],\n\t\"Further hits\": [\n";
            for (int64_t j = top, i = top; i < cnt && j < p.beasts; ++i) {
                o += j == top ? "\t{\n" : ",\n\t{\n"; obj(o, res[i], best, true);
                if (before != std::get<1>(res[i])) { before = std::get<1>(res[i]); ++j; }
            }
            o += "\n\t]\n}";
        } break;
        case Params::JsonL: {
            o += "{ \"Read number\": "; itoa(number, o); o += ", \"Specifier from input file\": \""; o += name; o += "\", \"Length\": "; itoa(len, o); o += ", \"Top hits\": [";
            for (int64_t i = 0; i < top; ++i) { o += i == 0 ? "{" : ",{"; obj(o, res[i], best, false); }
            o += "], \"Further hits\": [";
            for (int64_t j = top, i = top; i < cnt && j < p.beasts; ++i) {
                o += j == top ? "{" : ", {"; obj(o, res[i], best, false);
                if (before != std::get<1>(res[i])) { before = std::get<1>(res[i]); ++j; }
            }
            o += "] }\n";
        } break;
        case Params::Kraken: {
            o += "C\t"; o += name; o += "\t"; itoa(c.taxids[std::get<0>(res[0])], o); o += "\t"; itoa(len, o); o += "\t";
            for (int64_t i = 0; i < top; ++i) { itoa(c.taxids[std::get<0>(res[i])], o); o += ":"; dtoa(std::get<1>(res[i]), o); o += " "; }
            for (int64_t j = top, i = top; i < cnt && j < p.beasts; ++i) {
                itoa(c.taxids[std::get<0>(res[i])], o); o += ":"; dtoa(std::get<1>(res[i]), o); o += " ";
                if (before != std::get<1>(res[i])) { before = std::get<1>(res[i]); ++j; }
            }
            o += "\n";
        } break;
        }
    }
};

static void writeProfile(const string &path, const Params &p, const Content &c, const vector<double> &all, const vector<uint64_t> &uniq,
                         const vector<uint64_t> &total, const vector<uint64_t> &freqAll, uint64_t nKmers, uint64_t nReads)
{
    const int nK = p.kHigh - p.kLow + 1;
    const size_t nT = c.names.size();
    vector<uint64_t> sumU(nK, 0); vector<double> sumA(nK, 0.0);
    struct Row { string name; vector<std::pair<double, uint64_t>> v; uint32_t tid; size_t tix; };
    vector<Row> rows(nT, Row{"", vector<std::pair<double, uint64_t>>(nK, {0.0, 0}), 0, 0});
    for (size_t t = 1; t < nT; ++t) {
        Row r{c.names[t], vector<std::pair<double, uint64_t>>(nK), c.taxids[t], t};
        std::replace(r.name.begin(), r.name.end(), ',', ' ');
        for (int l = 0; l < nK; ++l) { r.v[l] = {all[l * nT + t], uniq[l * nT + t]}; sumU[l] += uniq[l * nT + t]; sumA[l] += all[l * nT + t]; }
        rows[t] = r;
    }
    std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) {
        for (size_t i = 0; i < a.v.size(); ++i) { if (a.v[i].second == b.v[i].second) continue; return a.v[i].second > b.v[i].second; }
        return false; });
    const uint64_t fm = p.frames == 1 ? 1 : ((p.frames == 6 && !p.protein) ? 6 : 3);   // Compare.hpp:3500
    vector<uint64_t> garbage(nK, 0);
    for (int i = p.kHigh - p.kLow, j = 0; i > 0; --i, ++j) garbage[j] = nReads * fm * i;
    std::ofstream f(path);
    if (!f) throw std::runtime_error("Profile file couldn't be opened for writing!");
    std::ostringstream body;
    f << "#taxID,Name";
    for (const char *title : {"Unique counts k=", "Unique rel. freq. k=", "Non-unique counts k=", "Non-unique rel. freq. k=", "Overall rel. freq. k=", "Overall unique rel. freq. k="})
        for (int l = 0; l < nK; ++l) f << "," << title << p.kHigh - l;
    if (p.coverage)                                                     // Compare.hpp:3574-3581
        for (const char *title : {"Special Counts k=", "Genome Coverage k="})
            for (int l = 0; l < nK; ++l) f << "," << title << p.kHigh - l;
    f << "\n";
    vector<double> ident(nK, 0), uident(nK, 0);
    for (const Row &r : rows) {
        if (!(r.v[nK - 1].first > 0)) continue;
        body << r.tid << "," << r.name;
        for (int l = 0; l < nK; ++l) body << "," << r.v[l].second;
        for (int l = 0; l < nK; ++l) { if (r.v[l].second == 0) body << "," << 0.0; else body << "," << static_cast<double>(r.v[l].second) / sumU[l]; }
        for (int l = 0; l < nK; ++l) body << "," << r.v[l].first;
        for (int l = 0; l < nK; ++l) { if (r.v[l].first == 0) body << "," << 0.0; else body << "," << r.v[l].first / sumA[l]; }
        for (int l = 0; l < nK; ++l) { ident[l] += r.v[l].first; body << "," << r.v[l].first / (nKmers - garbage[l]); }
        for (int l = 0; l < nK; ++l) { uident[l] += r.v[l].second; body << "," << static_cast<double>(r.v[l].second) / (nKmers - garbage[l]); }
        if (p.coverage) {                                               // Compare.hpp:3627-3637
            for (int l = 0; l < nK; ++l) body << "," << total[l * nT + r.tix];
            for (int l = 0; l < nK; ++l) body << "," << static_cast<double>(total[l * nT + r.tix]) / freqAll[l * nT + r.tix];
        }
        body << "\n";
    }
    f << "0,not identified";
    for (int l = 0; l < nK * 4; ++l) f << "," << 0.0;
    for (int l = 0; l < nK; ++l) f << "," << (double(nKmers) - double(garbage[l]) - ident[l]) / (double(nKmers) - double(garbage[l]));
    for (int l = 0; l < nK; ++l) f << "," << (double(nKmers) - double(garbage[l]) - uident[l]) / (double(nKmers) - double(garbage[l]));
    if (p.coverage) for (int l = 0; l < nK * 2; ++l) f << "," << 0.0;
    f << "\n" << body.str();
}

// kASA::setCodonTable (kASA.hpp:579-615): the 64 codons of table `id` of an NCBI gc.prt file written over the built-in table
static vector<uint8_t> codonTableFromFile(const string &path, const string &id)
{
    vector<uint8_t> lut(366);
    if (kasa_builtin_codon_table(lut.data())) throwLast();
    std::ifstream f(path);
    string line; bool found = false;
    while (std::getline(f, line)) if (line.find("  id " + id + " ,") != string::npos) { found = true; break; }
    if (!found) { std::cerr << "WARNING: codon table not found in file. Using built-in." << std::endl; return lut; }
    string aa, dummy, b1, b2, b3;
    std::getline(f, aa); std::getline(f, dummy); std::getline(f, b1); std::getline(f, b2); std::getline(f, b3);
    size_t pa = aa.find_first_of('"') + 1, pb = b1.find_first_of("TGCA");
    for (; pb < b1.size() && pb < b2.size() && pb < b3.size() && pa < aa.size(); ++pb, ++pa) {
        const int idx = ((b1[pb] & 14) << 5) | ((b2[pb] & 14) << 2) | ((b3[pb] & 14) >> 1);
        if (idx < 366) lut[(size_t)idx] = (uint8_t)(((aa[pa] == '*') ? '[' : aa[pa]) & 31);
    }
    return lut;
}

// Compare::filter (Compare.hpp:2448-2596): the input is read again and every record goes to <clean>.fast[aq] or
// <contaminants>.fast[aq] ("_1"/"_2" before the extension for paired input; "_" = that side is not written).
static string slurp(const string &path)
{
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) throw std::runtime_error("Input file not found");
    string data; vector<char> buf(1 << 22); int n;
    while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) data.append(buf.data(), (size_t)n);
    gzclose(g);
    return data;
}

static void filterReads(const Params &p, const vector<uint64_t> &flagged)
{
    const bool paired = !p.input2.empty();
    vector<string> data{slurp(p.input)};
    if (paired) data.push_back(slurp(p.input2));
    const bool fasta = !data[0].empty() && data[0][0] == '>';
    const string ext = string(fasta ? ".fasta" : ".fastq") + (p.gzipOut ? ".gz" : "");      // Compare.hpp:2454-2455
    // one side (clean or contaminants): a text per input file, written -- plain or through zlib (--gzip, as the reference's
    // ogzstream does) -- when the input has been walked
    struct Side {
        vector<string> names, text; bool on = false;
        void flush(bool gz, const char *what)
        {
            for (size_t i = 0; i < names.size(); ++i) {
                if (gz) {
                    gzFile g = gzopen(names[i].c_str(), "wb");
                    if (!g) throw std::runtime_error(string(what) + " output files could not be opened for writing, did you use the correct path?");
                    for (size_t a = 0; a < text[i].size();) { const unsigned n = (unsigned)std::min<size_t>(text[i].size() - a, 1u << 30); if (gzwrite(g, text[i].data() + a, n) <= 0) break; a += n; }
                    gzclose(g);
                } else {
                    std::ofstream f(names[i], std::ios::binary);
                    if (f.fail()) throw std::runtime_error(string(what) + " output files could not be opened for writing, did you use the correct path?");
                    f.write(text[i].data(), (std::streamsize)text[i].size());
                }
            }
        }
    };
    auto openSide = [&](const string &prefix) {
        Side sd;
        if (prefix == "_") return sd;
        sd.on = true;
        for (size_t i = 0; i < data.size(); ++i) { sd.names.push_back(prefix + (paired ? (i ? "_2" : "_1") : "") + ext); sd.text.emplace_back(); }
        return sd;
    };
    Side clean = openSide(p.filterClean), cont = openSide(p.filterCont);
    auto finish = [&]() { clean.flush(p.gzipOut, "Filtered"); cont.flush(p.gzipOut, "Contaminants"); };
    if (flagged.empty() && clean.on) { for (size_t i = 0; i < data.size(); ++i) clean.text[i] = data[i]; finish(); return; }
    vector<vector<std::pair<size_t, size_t>>> lines(data.size());
    for (size_t f = 0; f < data.size(); ++f)
        for (size_t a = 0; a < data[f].size();) { size_t b = data[f].find('\n', a); if (b == string::npos) b = data[f].size(); lines[f].emplace_back(a, b); a = b + 1; }
    auto line = [&](size_t f, size_t i) { return i < lines[f].size() ? data[f].substr(lines[f][i].first, lines[f][i].second - lines[f][i].first) : string(); };
    uint64_t rid = 0; size_t fi = 0;
    Side *target = &clean;
    const size_t n = lines[0].size();
    if (fasta) {
        for (size_t i = 0; i < n; ++i) {
            const string l1 = line(0, i);
            if (l1.empty()) continue;
            if (l1[0] == '>') { const bool hit = fi < flagged.size() && rid == flagged[fi]; target = hit ? &cont : &clean; if (hit) ++fi; ++rid; }
            if (target->on) { target->text[0] += l1; target->text[0] += "\n"; if (paired) { target->text[1] += line(1, i); target->text[1] += "\n"; } }
        }
    } else {
        for (size_t i = 0; i < n; i += 4) {
            if (line(0, i).empty()) continue;
            const bool hit = fi < flagged.size() && rid == flagged[fi];
            target = hit ? &cont : &clean; if (hit) ++fi; ++rid;
            if (!target->on) continue;
            for (size_t f = 0; f < data.size(); ++f) for (size_t k = 0; k < 4; ++k) { target->text[f] += line(f, i + k); target->text[f] += "\n"; }
        }
    }
    finish();
}

// ---------------------------------------------------------------------------------------------------
// the batch pipeline of one input file
// ---------------------------------------------------------------------------------------------------
// A host buffer that crosses PCIe: page-locked (kasa_host_alloc) when it can be had -- transfers from pageable memory are
// staged by the runtime at a fraction of the link rate.
template <class T> struct PcieBuf {
    T *p = nullptr; size_t n = 0; bool pinned = false;
    PcieBuf() {}
    PcieBuf(const PcieBuf &) = delete;
    PcieBuf &operator=(const PcieBuf &) = delete;
    ~PcieBuf() { release(); }
    void release() { if (p) { if (pinned) kasa_host_free(p); else std::free(p); } p = nullptr; n = 0; capacity = 0; }
    size_t capacity = 0;
    // grow-only: page-locking memory costs more than the transfer it speeds up, so a worker keeps its buffers over the batches
    void resize(size_t count)
    {
        if (count <= capacity && p) { n = count; return; }
        release();
        count += count / 8;
        capacity = count;
        n = count;
        const size_t bytes = std::max<size_t>(1, count * sizeof(T));
        p = (T *)kasa_host_alloc(bytes);
        pinned = p != nullptr;
        if (!p) p = (T *)std::malloc(bytes);
        if (!p) throw std::bad_alloc();
    }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

struct SplitCarry;
struct Batch {
    uint64_t id = 0, firstRead = 0;
    SplitCarry *carry = nullptr;
    ReadSet rs;                                   // the batch's reads (offsets start at 0)
    vector<uint32_t> segRead;                     // paired-end: read of every sequence
    vector<uint64_t> flagged;                     // --filter: read numbers of contaminants
    uint64_t kmers = 0;
    uint32_t flaggedByDevice = 0;                 // reads kasa_batch_rank handed back to the host's std::sort
    bool done = false;
    bool head = false, tail = false;              // read 0 goes on from the batch before / the last read goes on in the next batch (pieces, below)
};

// What an unfinished read has scored so far (Compare::saveResults' vSavedScores, Compare.hpp:2324-2443): left by the batch
// that ends inside a read for the batch that goes on with it -- which may be another worker's, hence the lock.
struct SplitCarry {
    std::mutex mu; std::condition_variable cv;
    std::map<uint64_t, vector<std::pair<uint32_t, float>>> after;   // batch id -> what waits behind it
    bool abandoned = false;
    void put(uint64_t id, vector<std::pair<uint32_t, float>> v) { { std::lock_guard<std::mutex> lk(mu); after[id] = std::move(v); } cv.notify_all(); }
    vector<std::pair<uint32_t, float>> take(uint64_t id)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return abandoned || after.count(id); });
        if (abandoned) throw std::runtime_error("another batch failed");
        auto v = std::move(after[id]); after.erase(id);
        return v;
    }
    void abandon() { { std::lock_guard<std::mutex> lk(mu); abandoned = true; } cv.notify_all(); }
    // taxa ascending on both sides; a taxon on both sides gets the float sum (Compare.hpp:2347-2360)
    static vector<std::pair<uint32_t, float>> merge(const vector<std::pair<uint32_t, float>> &a, const uint32_t *tax, const float *sc, uint64_t n)
    {
        vector<std::pair<uint32_t, float>> o; o.reserve(a.size() + n);
        size_t i = 0; uint64_t j = 0;
        while (i < a.size() || j < n) {
            if (j == n || (i < a.size() && a[i].first < tax[j])) o.push_back(a[i++]);
            else if (i == a.size() || tax[j] < a[i].first) { o.emplace_back(tax[j], sc[j]); ++j; }
            else { o.emplace_back(tax[j], a[i].second + sc[j]); ++i; ++j; }
        }
        return o;
    }
};

// --- sequences the reference reads in pieces ---------------------------------------------------------------------------
// Read::readFileAndGenerateInfos (Read.hpp:371-600) ends a piece after the getChunk call with which the k-mers of what it has
// read of the record so far would take more than 100 MiB of the input vector (FASTA: line feeds count as letters there); what
// is left when the record ends is the last piece, possibly empty.  -> letters before every cut, and what every piece adds
// to "Length" (letters + line feeds, Read.hpp:723-731).
struct Pieces { vector<int64_t> cut, add; };
static Pieces readerPieces(const ReadSet::LongRec &lr, bool fasta, bool protein, int frames, int64_t K, bool coherence)
{
    const int mode = protein ? 2 : (frames == 1 ? 1 : 0), strands = (frames == 6 && !protein) ? 2 : 1;
    const int64_t elem = coherence ? (K > 12 ? 40 : 32) : (K > 12 ? 32 : 24);                  // InputType::sizeOf, MetaHeader.h:221-223
    const int64_t mult = elem * ((strands == 2 && mode != 2) ? 2 : 1), limit = kPieceBytes;
    auto count = [&](int64_t len) -> int64_t {                                                  // Read.hpp:36-57
        if (mode == 2) return len > K + 1 ? len - K + 1 : 0;
        if (mode == 1) return len / 3 > K + 1 ? len / 3 - K + 1 : 0;
        return len > 3 * K + 1 ? len - 3 * K + 1 : 0;
    };
    Pieces pc; pc.cut.push_back(0);
    int64_t chars = 0, letters = 0, total = 0, totalAtCut = 0;
    for (size_t i = 0; i < lr.letters.size(); ++i) {
        const int64_t l = lr.letters[i], f = lr.feed[i];
        letters += l; total += l + f;
        if (l > 0) chars += l + (fasta ? f : 0);                                                // (a call without text is not counted: Read.hpp:394,445)
        if (l > 0 && count(chars) * mult > limit) { pc.cut.push_back(letters); pc.add.push_back(total - totalAtCut); totalAtCut = total; chars = 0; }
    }
    pc.cut.push_back(letters); pc.add.push_back(total - totalAtCut);
    return pc;
}

// Cuts the reads of one input into batches.  With per-read output the batches are the reference's own: per-read scores
// are float sums whose order depends on the reads sharing a batch, so the input is cut exactly where `kASA identify -m`
// cuts it (kasa_refbatch_*: Compare.hpp:2803-2818,3129-3132; Read.hpp:612-630,1147,1165-1195).  A profile-only run
// (boundaries do not matter there) and a reference batch that does not fit the device are cut by free HBM instead.
struct Batcher {
    const Params &p; const IndexFiles &ixf;
    bool wantRows, paired;
    std::unique_ptr<ChunkReader> reader;          // single-end input is streamed
    ReadSet pending; size_t pendPos = 0;          // parsed reads not yet handed out
    vector<uint32_t> pendSeg;                     // paired-end: the whole (merged) input sits in `pending`
    uint64_t nextRead = 0, nextId = 0;
    int64_t refBudget = 0; bool useRef = false, firstBatch = true, warned = false;
    uint64_t maxKmersPerBatch;
    double parseSeconds = 0;
    bool protein = false;
    // pieces (nextPieced): the read at pendPos has handed out `midPiece` of its pieces; "Length" as the reader's calls count it
    bool midRead = false; size_t midPiece = 0;
    int64_t carriedLen = 0;                       // strTransfer::lengthOfDNA (Read.hpp:1181)

    Batcher(const Params &pp, const IndexFiles &f, bool rows, uint64_t maxKmers) : p(pp), ixf(f), wantRows(rows), paired(!pp.input2.empty()), maxKmersPerBatch(maxKmers)
    {
        const auto t0 = std::chrono::steady_clock::now();
        if (paired) {
            // paired-end (Read.hpp:834-1049): mate r of both files forms read r; the two sequences stay separate (no k-mer
            // spans the junction) but score into one row; specifier = both names, length = the sum
            ReadSet r1 = readInput(p.input, p.verbose, p.threads), r2 = readInput(p.input2, false, p.threads);
            if (r2.size() != r1.size()) throw std::runtime_error("The paired-end files hold different numbers of reads");
            pending.protein = r1.protein;
            for (size_t r = 0; r < r1.size(); ++r) {
                for (const ReadSet *x : {&r1, &r2}) {
                    pending.bases.append(x->bases.data() + x->off[r], (size_t)(x->off[r + 1] - x->off[r]));
                    pending.off.push_back((int64_t)pending.bases.size());
                    pendSeg.push_back((uint32_t)r);
                    pending.nameBlob.append(x->nameBlob.data() + x->nameOff[r], x->nameLen(r));
                }
                pending.nameOff.push_back(pending.nameBlob.size());
                pending.lengths.push_back(r1.lengths[r] + r2.lengths[r]);
            }
            protein = pending.protein;
        } else {
            reader.reset(new ChunkReader(p.input));
            refill();
            protein = reader->protein;
        }
        parseSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (wantRows && !getenv("KASA_MAX_BATCH_KMERS")) {
            kasa_refbatch_params bp{p.memoryGiB, p.refThreads, p.ram ? 1 : 0, p.kHigh, p.kLow, ixf.recBytes, ixf.nRec, ixf.tp.data(), ixf.tp.size(),
                                    ixf.content.taxids.data(), (uint32_t)ixf.content.taxids.size(), ixf.nameBytes, p.mode == "identify_multiple" ? 1 : 0};
            if (kasa_refbatch_budget(&bp, &refBudget)) throw std::runtime_error("batch budget could not be computed");
            useRef = true;
        }
    }
    size_t seqPerRead() const { return paired ? 2 : 1; }
    size_t pendingReads() const { return pending.size() - pendPos; }
    vector<ReadSet> parts;                         // the parse threads' runs (kept: their pages are touched once)
    void refill()                                  // parse the next chunk of the file behind the pending reads
    {
        if (!reader) return;
        ++pendingGen;
        if (pendPos > 0 && pendPos == pending.size()) { pending.clear(); pendPos = 0; }
        const char *chunk = nullptr; size_t chunkBytes = 0;
        if (!reader->next(chunk, chunkBytes, p.verbose)) return;
        if (pendPos > 0) {                          // drop what was handed out, keep the rest
            ReadSet rest = pending.slice(pendPos, pending.size(), 1, p.threads);
            pending = std::move(rest); pendPos = 0;
        }
        if (pending.size() == 0 && reader->fileSize > 0) {
            // room for what is still to come (address space only: untouched pages cost nothing), so that a batch that
            // takes many chunks never moves what it already holds
            const size_t left = (size_t)(reader->fileSize - std::min(reader->fileSize, reader->filePos)) + chunkBytes;
            pending.bases.reserve(left / 2 + left / 8 + 1024); pending.nameBlob.reserve(left / 8 + 1024);
            pending.nameOff.reserve(left / 160 + 17); pending.lengths.reserve(left / 160 + 16); pending.off.reserve(left / 160 + 17);
        }
        parsePiece(chunk, chunkBytes, reader->fasta, p.threads, 1u << 20, pending, parts, reader->chunkStart);
    }
    // What the first batch will need on the device, from the reads parsed so far and the size of the file: the device buffers
    // are allocated while the rest of the input is parsed (kasa_ctx_reserve).
    void estimateFirstBatch(uint64_t &nQueries, uint64_t &nBases) const
    {
        nQueries = nBases = 0;
        const size_t have = pending.size() - pendPos;
        if (have == 0) return;
        const int mode = protein ? 2 : (p.frames == 1 ? 1 : 0), strands = (p.frames == 6 && !protein) ? 2 : 1;
        const size_t spr = seqPerRead(), sample = std::min<size_t>(have, 65536);
        const uint32_t nTaxa = (uint32_t)ixf.content.taxids.size();
        double bases = 0, cost = 0;
        for (size_t r = pendPos; r < pendPos + sample; ++r) {
            if (useRef) cost += (double)kasa_refbatch_read_overhead((int64_t)pending.nameLen(r), nTaxa, p.coherence ? 1 : 0);
            for (size_t q = 0; q < spr; ++q) {
                const int64_t l = pending.off[r * spr + q + 1] - pending.off[r * spr + q];
                bases += (double)l;
                if (useRef) cost += (double)kasa_refbatch_sequence_cost(p.K, p.kLow, mode, strands, l, p.coherence ? 1 : 0);
            }
        }
        bases /= (double)sample; cost /= (double)sample;
        double reads = (double)have;                                         // reads of the whole input, by the share of the file parsed
        if (reader && reader->fileSize > 0 && reader->filePos > 0 && !reader->g) reads *= std::max(1.0, (double)reader->fileSize / (double)reader->filePos);
        if (useRef && cost > 0) reads = std::min(reads, std::max(1.0, ((double)refBudget - 100.0 * 1024 * 1024) / cost));
        double q = reads * (bases + 8.0 * spr) * strands;
        if (q > (double)maxKmersPerBatch) { reads *= (double)maxKmersPerBatch / q; q = (double)maxKmersPerBatch; }
        nQueries = (uint64_t)(q * 1.01); nBases = (uint64_t)(reads * bases * 1.01);
    }
    // the next batch; false at the end of the input
    Pieces piecesOf(const ReadSet::LongRec &lr, bool fasta) const { return readerPieces(lr, fasta, protein, p.frames, p.K, p.coherence); }
    std::map<size_t, Pieces> pieceCache;           // read (index in `pending`) -> its pieces, for records with more than one
    uint64_t pendingGen = 0, cachedGen = ~0ull;    // `pending` changes (reads come, handed-out ones go: other numbers) -> the cache is made again
    const Pieces *piecesOfRead(size_t r)
    {
        if (cachedGen != pendingGen) {
            pieceCache.clear();
            for (const ReadSet::LongRec &lr : pending.longRecs) { Pieces pc = piecesOf(lr, pending.fasta); if (pc.cut.size() > 2) pieceCache.emplace(lr.read, std::move(pc)); }
            cachedGen = pendingGen;
        }
        auto it = pieceCache.find(r);
        return it == pieceCache.end() ? nullptr : &it->second;
    }
    bool pendingHasPieces() { (void)piecesOfRead(0); for (auto &kv : pieceCache) if (kv.first >= pendPos) return true; return false; }
    // A batch over an input with such records: the unit is the piece.  A piece takes its k-mers and its text from the budget
    // when it is read, the read's own share goes with its last piece (Read.hpp:1157-1194); the batch may end between two
    // pieces of a read (strTransfer, Read.hpp:343-356) -- the read is then the unfinished last one of this batch and read 0
    // of the next (Batch::tail / head; runBatch keeps its scores).  Every piece but a read's first starts with the last 3K-1
    // letters of the text before it (Read.hpp:678-697,738-741).
    bool nextPieced(Batch &b, std::chrono::steady_clock::time_point t0)
    {
        if (p.coherence) throw std::runtime_error("--coherence over sequences long enough for kASA to read them in pieces is not supported");
        const int mode = protein ? 2 : (p.frames == 1 ? 1 : 0), strands = (p.frames == 6 && !protein) ? 2 : 1;
        const uint32_t nTaxa = (uint32_t)ixf.content.taxids.size();
        const int64_t over = (protein ? p.K : 3 * p.K) - 1;
        int64_t left = refBudget;
        if (useRef && !firstBatch && refBudget - (int64_t)(refBudget * 0.001) > 0) left -= (int64_t)(refBudget * 0.001);
        struct Item { size_t r; size_t piece, nPieces; int64_t from, to; };     // letters [from, to) of read r (from: with the overhang)
        vector<Item> items;
        uint64_t est = 0;
        bool full = false, deviceFull = false;
        size_t r = pendPos, piece = midRead ? midPiece : 0;
        b.head = midRead;
        while (!full) {
            if (r == pending.size()) {
                if (!items.empty() || midRead) {
                    // (refill drops the reads before pendPos: the items of this batch must stay, so the chunk goes behind them)
                    const size_t before = pending.size();
                    const char *chunk = nullptr; size_t chunkBytes = 0;
                    if (!reader || !reader->next(chunk, chunkBytes, p.verbose)) break;
                    parsePiece(chunk, chunkBytes, reader->fasta, p.threads, 1u << 20, pending, parts, reader->chunkStart);
                    ++pendingGen;
                    if (pending.size() == before) break;
                } else {
                    const size_t before = pending.size() - pendPos;
                    refill();
                    r = pendPos;
                    if (pending.size() - pendPos == before) break;
                }
            }
            ScopedTimer tmForm(g_ht.form);
            for (; r < pending.size() && !full; ++r, piece = 0) {
                const Pieces *pc = piecesOfRead(r);
                const int64_t len = pending.off[r + 1] - pending.off[r];
                const size_t nP = pc ? pc->cut.size() - 1 : 1;
                int64_t textLen = 0;                                             // of the piece before (for the overhang)
                for (size_t i = 0; i < piece; ++i) textLen = std::min(over, textLen) + pc->cut[i + 1] - pc->cut[i];
                for (; piece < nP; ++piece) {
                    if (useRef && left <= 100ll * 1024 * 1024 && !items.empty()) { full = true; break; }     // Read.hpp:1147
                    const int64_t c0 = pc ? pc->cut[piece] : 0, c1 = pc ? pc->cut[piece + 1] : len;
                    const int64_t keep = piece ? std::min(over, textLen) : 0;
                    const int64_t tl = keep + c1 - c0;
                    int64_t cost = 0;
                    if (useRef) {
                        cost = kasa_refbatch_sequence_cost(p.K, p.kLow, mode, strands, tl, 0);
                        if (piece + 1 == nP) cost += kasa_refbatch_read_overhead((int64_t)pending.nameLen(r), nTaxa, 0);
                    }
                    const uint64_t k = (uint64_t)(tl + 64) * (uint64_t)strands;
                    if (!items.empty() && est + k > maxKmersPerBatch) { deviceFull = true; full = true; break; }
                    est += k; left -= cost;
                    items.push_back({r, piece, nP, c0 - keep, c1});
                    textLen = tl;
                }
                if (full) break;
            }
        }
        if (items.empty()) { parseSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); return false; }
        {
            ScopedTimer tmForm(g_ht.form);
            ReadSet &o = b.rs;
            o.protein = pending.protein; o.fasta = pending.fasta;
            int64_t runLen = carriedLen;                                         // iLengthOfRead starts from what the batches before have read of the read (Read.hpp:1117)
            size_t lastR = ~(size_t)0; uint32_t local = 0;
            for (const Item &it : items) {
                if (it.r != lastR) {
                    if (lastR != ~(size_t)0) ++local;
                    lastR = it.r;
                    o.nameBlob.append(pending.nameBlob.data() + pending.nameOff[it.r], pending.nameLen(it.r));
                    o.nameOff.push_back(o.nameBlob.size());
                    o.lengths.push_back(pending.lengths[it.r]);
                }
                o.bases.append(pending.bases.data() + pending.off[it.r] + it.from, (size_t)(it.to - it.from));
                o.off.push_back((int64_t)o.bases.size());
                b.segRead.push_back(local);
                if (it.nPieces > 1) {
                    const Pieces *pc = piecesOfRead(it.r);
                    runLen += pc->add[it.piece];
                    if (it.piece + 1 == it.nPieces) { o.lengths[local] = (uint32_t)runLen; runLen = 0; carriedLen = 0; }
                    else carriedLen += runLen;                                   // Read.hpp:1181: grows by the running length at every unfinished piece
                } else runLen = 0;
            }
            const Item &lastItem = items.back();
            b.tail = lastItem.piece + 1 < lastItem.nPieces;
            midRead = b.tail; midPiece = b.tail ? lastItem.piece + 1 : 0;
            pendPos = b.tail ? lastItem.r : lastItem.r + 1;
            const uint64_t nLocal = (uint64_t)local + 1;
            b.firstRead = nextRead;
            nextRead += nLocal - (b.tail ? 1 : 0);
        }
        if (useRef && deviceFull && !p.allowDeviceSplit)
            throw std::runtime_error("a batch of the reference's size (-m " + std::to_string(p.memoryGiB) + ") does not fit the device; use a smaller -m, more devices' memory, or --allow-device-split");
        firstBatch = false;
        parseSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return true;
    }
    bool next(Batch &b)
    {
        const auto t0 = std::chrono::steady_clock::now();
        b = Batch();
        b.id = nextId++; b.firstRead = nextRead;
        if (!paired && (midRead || pendingHasPieces())) return nextPieced(b, t0);
        if (paired && !pending.longRecs.empty() && pendingHasPieces())
            throw std::runtime_error("a paired-end input with sequences long enough for kASA to read them in pieces is not supported");
        const int mode = protein ? 2 : (p.frames == 1 ? 1 : 0), strands = (p.frames == 6 && !protein) ? 2 : 1;
        const size_t spr = seqPerRead();
        int64_t left = refBudget;
        if (useRef && !firstBatch && refBudget - (int64_t)(refBudget * 0.001) > 0) left -= (int64_t)(refBudget * 0.001);   // Compare.hpp:3129-3132
        uint64_t est = 0, n = 0;
        bool deviceFull = false;
        bool full = false;
        const uint32_t nTaxa = (uint32_t)ixf.content.taxids.size();
        // The batch's reads are pending[pendPos, r): counted here read by read with the reference's arithmetic; chunks are
        // parsed behind them as long as the batch has room, and only then the reads change hands -- in one piece.
        size_t r = pendPos;
        while (!full) {
            if (r == pending.size()) {
                const size_t before = pending.size() - pendPos;
                const size_t keep = r - pendPos;
                refill();                                                   // (may drop the reads before pendPos)
                r = pendPos + keep;
                if (pending.size() - pendPos == before) break;             // end of the input
                if (!paired && pendingHasPieces()) return nextPieced(b, t0);   // a record the reference reads in pieces has come in: the batch is formed piece by piece
            }
            ScopedTimer tmForm(g_ht.form);
            for (; r < pending.size(); ++r) {
                if (useRef && left <= 100ll * 1024 * 1024 && n > 0) { full = true; break; }                        // Read.hpp:1147
                uint64_t len = 0;
                int64_t cost = useRef ? kasa_refbatch_read_overhead((int64_t)pending.nameLen(r), nTaxa, p.coherence ? 1 : 0) : 0;
                for (size_t q = 0; q < spr; ++q) {
                    const int64_t l = pending.off[r * spr + q + 1] - pending.off[r * spr + q];
                    len += (uint64_t)l;
                    if (useRef) cost += kasa_refbatch_sequence_cost(p.K, p.kLow, mode, strands, l, p.coherence ? 1 : 0);
                }
                const uint64_t k = (len + 64 * spr) * (uint64_t)strands;
                if (n > 0 && est + k > maxKmersPerBatch) { deviceFull = true; full = true; break; }
                est += k; left -= cost; ++n;
            }
        }
        {
            ScopedTimer tmForm(g_ht.form);
            const size_t first = pendPos, m = r - first;
            if (m > 0 && first == 0 && r == pending.size() && !paired) {
                b.rs = std::move(pending);                              // everything that is pending: nothing is copied
                pending = ReadSet(); pendPos = 0; ++pendingGen;
            } else if (m > 0) {
                b.rs = pending.slice(first, r, spr, p.threads);
                if (paired) { b.segRead.resize(2 * m); for (size_t x = 0; x < m; ++x) b.segRead[2 * x] = b.segRead[2 * x + 1] = (uint32_t)x; }
                pendPos = r;
            }
        }
        if (useRef && deviceFull && !p.allowDeviceSplit)
            // per-read scores are float sums whose order depends on the reads that share a batch (Compare.hpp:528-530): a batch cut
            // elsewhere than kASA cuts it is not kASA's result -- an error, not a warning with different digits
            throw std::runtime_error("a batch of the reference's size (-m " + std::to_string(p.memoryGiB) + ") does not fit the device; use a smaller -m (kASA's batches shrink with it), "
                                     "more devices' memory, or --allow-device-split (the batch is cut where the device is full: per-read scores may differ from kASA's in their last digit)");
        if (useRef && deviceFull && !warned) {
            std::cerr << "WARNING: a batch of the reference's size does not fit the device; it is cut (--allow-device-split): per-read scores may differ in their last digit." << std::endl;
            warned = true;
        }
        firstBatch = false;
        nextRead += n;
        parseSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return n > 0;
    }
};

// The per-read file.  A slab of text (the reads [32768 s, 32768 (s + 1)) of a batch) gets its place in the file as soon as
// the sizes of all slabs before it are known, and is written there by the thread that formatted it: formatting and writing
// overlap, several threads fill the page cache at once (one thread does 2-3 GB/s; 10 M reads are 5.5 GB of JSON lines), and
// no slab is copied into a batch-sized string first.  Slabs that arrive early wait (text of a later batch on another device).
struct OrderedOut {
    int fd = -1; off_t pos = 0;
    std::mutex mu;
    uint64_t curBatch = 0; size_t curSlab = 0;                  // the next slab to be placed
    // a slab: text the formatting thread made (owned), or the piece a device wrote (a view into the buffer it arrived in; `done`
    // tells the buffer's owner when it may be used again)
    struct Item { string text; const char *view = nullptr; size_t n = 0; std::shared_ptr<std::promise<void>> done; size_t size() const { return view ? n : text.size(); } };
    std::map<std::pair<uint64_t, size_t>, Item> parked;
    std::map<uint64_t, size_t> slabsOf;
    // views are written by ONE thread of their own: new tmpfs pages take 7 GB/s from one writer and 3.5 GB/s from two or
    // sixteen (tools/hostio_probe.cpp), and the device worker goes on with its next batch meanwhile
    std::thread writer; std::mutex wmu; std::condition_variable wcv; std::deque<std::pair<off_t, Item>> wq; bool wStop = false; size_t wBusy = 0;
    std::exception_ptr wErr;
    bool abandoned = false;                                     // a batch failed: nothing more is placed, nobody waits for a place
    ~OrderedOut() { stopWriter(); if (fd >= 0) ::close(fd); }
    static void writeAt(int fd, const char *d, size_t n, off_t at)
    {
        while (n) { const ssize_t w = ::pwrite(fd, d, std::min<size_t>(n, (size_t)1 << 30), at); if (w <= 0) throw std::runtime_error("Readwise output file could not be written!"); d += w; n -= (size_t)w; at += w; }
    }
    void put(const string &t) { writeAt(fd, t.data(), t.size(), pos); pos += (off_t)t.size(); }   // header / footer (nothing else in flight)
    void place(vector<std::pair<off_t, Item>> &todo)               // mu held: everything that is next in line gets its offset
    {
        for (;;) {
            auto sit = slabsOf.find(curBatch);
            if (sit != slabsOf.end() && curSlab >= sit->second) { slabsOf.erase(sit); ++curBatch; curSlab = 0; continue; }
            auto it = parked.find({curBatch, curSlab});
            if (it == parked.end()) return;
            todo.emplace_back(pos, std::move(it->second));
            pos += (off_t)todo.back().second.size();
            parked.erase(it);
            ++curSlab;
        }
    }
    void writerLoop()
    {
        for (;;) {
            std::pair<off_t, Item> job;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return wStop || !wq.empty(); });
                if (wq.empty()) return;
                job = std::move(wq.front()); wq.pop_front(); ++wBusy;
            }
            try {
                ScopedTimerMt tm(g_ht.write, g_ht.mu);
                writeAt(fd, job.second.view, job.second.n, job.first);
                job.second.done->set_value();
            } catch (...) { job.second.done->set_exception(std::current_exception()); std::lock_guard<std::mutex> lk(wmu); if (!wErr) wErr = std::current_exception(); }
            { std::lock_guard<std::mutex> lk(wmu); --wBusy; }
            wcv.notify_all();
        }
    }
    void dispatch(vector<std::pair<off_t, Item>> &todo)            // (no lock held) owned text: written here; views: handed to the writer
    {
        for (auto &w : todo) {
            if (w.second.view) {
                std::lock_guard<std::mutex> lk(wmu);
                if (!writer.joinable()) writer = std::thread([this] { writerLoop(); });
                wq.push_back(std::move(w));
                wcv.notify_all();
            } else { ScopedTimerMt tm(g_ht.write, g_ht.mu); writeAt(fd, w.second.text.data(), w.second.text.size(), w.first); }
        }
    }
    void abandon()                                                 // after a failure: what waits for its place is dropped, its owners are released
    {
        std::lock_guard<std::mutex> lk(mu);
        abandoned = true;
        for (auto &kv : parked) if (kv.second.done) kv.second.done->set_value();
        parked.clear();
    }
    void drainWriter()                                             // everything handed to the writer is in the file
    {
        std::unique_lock<std::mutex> lk(wmu);
        wcv.wait(lk, [&] { return wq.empty() && wBusy == 0; });
        if (wErr) { auto e = wErr; wErr = nullptr; std::rethrow_exception(e); }
    }
    void stopWriter()
    {
        { std::lock_guard<std::mutex> lk(wmu); wStop = true; }
        wcv.notify_all();
        if (writer.joinable()) writer.join();
    }
    void begin(uint64_t batch, size_t nSlabs)
    {
        if (fd < 0) return;
        vector<std::pair<off_t, Item>> todo;
        { std::lock_guard<std::mutex> lk(mu); slabsOf[batch] = nSlabs; place(todo); }
        dispatch(todo);
    }
    void submit(uint64_t batch, size_t slab, string &&t)
    {
        if (fd < 0) return;
        vector<std::pair<off_t, Item>> todo;
        { std::lock_guard<std::mutex> lk(mu); Item it; it.text = std::move(t); parked.emplace(std::make_pair(batch, slab), std::move(it)); place(todo); }
        dispatch(todo);
    }
    // A slab that the caller keeps (the buffer a piece of the device's text arrived in) until the returned future is ready.
    std::future<void> view(uint64_t batch, size_t slab, const char *d, size_t n)
    {
        auto done = std::make_shared<std::promise<void>>();
        std::future<void> f = done->get_future();
        if (fd < 0) { done->set_value(); return f; }
        vector<std::pair<off_t, Item>> todo;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (abandoned) { done->set_value(); return f; }
            Item it; it.view = d; it.n = n; it.done = done;
            parked.emplace(std::make_pair(batch, slab), std::move(it));
            place(todo);
        }
        dispatch(todo);
        return f;
    }
};

// Everything one device does for one batch: upload -> encode -> sort -> lookup/score on the device (the calls
// CompareWithLib_partialSort makes per batch, Compare.hpp:3107-3310), CSR back, ranking + text by all host threads (N2).
// what crosses PCIe for one worker (device): kept over its batches
struct WorkerBuffers {
    PcieBuf<uint32_t> meta; PcieBuf<Writer::DeviceHit> hits;
    PcieBuf<uint64_t> ro; PcieBuf<uint32_t> tx; PcieBuf<float> sc;
    // the device's text (kasa_batch_text) comes through a few small page-locked buffers (making one of 5.5 GB takes seconds):
    // the writer thread empties one while the next pieces arrive
    // ten of them hold the text of a reference-sized batch (-m 12: 550 MB), so the device can be a whole batch ahead of the
    // writer; they are made when the worker starts, while the input is still being parsed (page-locking takes 0.2 s per GB)
    static constexpr unsigned TEXT_SLOTS = 10;
    PcieBuf<char> text[TEXT_SLOTS]; std::future<void> written[TEXT_SLOTS]; unsigned turn = 0;
    size_t pieceBytes = (size_t)64 << 20;
    void prepareText()
    {
        if (const char *e = getenv("KASA_TEXT_PIECE")) pieceBytes = std::max<size_t>(1, (size_t)atoll(e));   // tests force small pieces
        for (auto &t : text) if (t.capacity < pieceBytes) t.resize(pieceBytes);
    }
    PcieBuf<uint8_t> flags;
    ~WorkerBuffers() { for (auto &f : written) if (f.valid()) f.wait(); }      // the writer still reads the buffers
};

static void runBatch(const Params &p, const IndexFiles &ixf, kasa_ctx *ctx, const vector<kasa_ctx *> &partCtx, Batch &b, bool wantRows, double &tDevice, double &tText, WorkerBuffers &wb, OrderedOut &out)
{
    const auto tDev = std::chrono::steady_clock::now();
    auto secondsSince = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const uint64_t nr = b.rs.size();
    {
        ScopedTimerMt tm(g_ht.upload, g_ht.mu);
        if (b.segRead.empty()) { if (kasa_batch_upload(ctx, b.rs.bases.data(), b.rs.off.data(), (int64_t)nr)) throwLast(); }
        else if (kasa_batch_upload_segments(ctx, b.rs.bases.data(), b.rs.off.data(), (int64_t)b.segRead.size(), b.segRead.data(), (int64_t)nr)) throwLast();
    }
    uint64_t nk = 0;
    {
        ScopedTimerMt tm(g_ht.compute, g_ht.mu);
        { ScopedTimerMt t2(g_ht.encode, g_ht.mu); if (kasa_batch_encode(ctx, &nk)) throwLast(); }
        { ScopedTimerMt t2(g_ht.sort, g_ht.mu); if (kasa_batch_sort_and_range(ctx, p.unique ? 1 : 0)) throwLast(); }
        ScopedTimerMt t2(g_ht.score, g_ht.mu);
        if (partCtx.empty()) { if (kasa_batch_lookup_score(ctx, wantRows, p.coverage)) throwLast(); }
        else {
            // a partitioned index: the sorted queries are cut at the partitions' first prefixes; every slice is grouped against
            // its partition by that partition's context (the slice is a batch of its own there: its first query opens a new
            // prefix range), the records come back in partition order = sorted order and are scored here (kasa_hip.h, C5 seam)
            const size_t nP = partCtx.size();
            const void *km = nullptr; uint64_t nq = 0;
            if (kasa_batch_queries_device(ctx, &km, &nq)) throwLast();
            vector<uint64_t> starts(nP + 1);
            if (kasa_batch_slice_starts(ctx, ixf.cuts.data(), (uint32_t)nP, starts.data())) throwLast();
            const size_t keyBytes = p.K > 12 ? 16 : 8;
            vector<const uint32_t *> recs(nP), pools(nP); vector<uint64_t> nRecW(nP), nPoolW(nP);
            for (size_t j = 0; j < nP; ++j) {
                if (kasa_batch_set_sorted_device(partCtx[j], (const char *)km + starts[j] * keyBytes, starts[j + 1] - starts[j])) throwLast();
                if (kasa_batch_group(partCtx[j], p.coverage)) throwLast();
                if (kasa_profile_absorb(ctx, partCtx[j])) throwLast();
                if (kasa_batch_records_device(partCtx[j], &recs[j], &nRecW[j], &pools[j], &nPoolW[j])) throwLast();
            }
            if (kasa_batch_records_import_device(ctx, (uint32_t)nP, recs.data(), nRecW.data(), pools.data(), nPoolW.data())) throwLast();
            if (kasa_batch_score(ctx, wantRows)) throwLast();
        }
    }
    b.kmers = nk;
    if (!wantRows) { tDevice += secondsSince(tDev); out.begin(b.id, 0); return; }
    vector<float> coherence;                                     // --coherence (Compare::postProcess, Compare.hpp:3317-3321)
    if (p.coherence) {
        coherence.assign(nr, 0.f);
        uint64_t throwsAt = ~0ull;
        if (kasa_batch_coherence(ctx, coherence.data(), &throwsAt)) throwLast();
        if (throwsAt != ~0ull)                                   // the reference's walk runs off the end of its vector here (vector::at)
            throw std::runtime_error("vector::_M_range_check: __n (which is " + std::to_string(throwsAt) + ") >= this->size() (which is " + std::to_string(throwsAt) + ")");
    }
    // Ranking on the device (kasa_batch_rank): the host supplies libm's denominators, one row per distinct read length,
    // and gets back only what the writer can print; the full rows come back when the device flags a read (a tie under
    // std::sort's unstable regime) or when there are too many distinct lengths for a table.
    PcieBuf<uint32_t> &meta = wb.meta; PcieBuf<Writer::DeviceHit> &hits = wb.hits;
    uint32_t nFlagged = 0;
    const bool split = b.head || b.tail;                         // a read of this batch is scored over several batches: its rows come to the host
    bool deviceRank = nr > 0 && !p.hostRank && !split;
    if (deviceRank) {
        std::map<uint32_t, uint32_t> classOf;
        { uint32_t last = ~0u; for (uint64_t r = 0; r < nr; ++r) if (b.rs.lengths[r] != last) { last = b.rs.lengths[r]; classOf.emplace(last, 0u); } }   // (runs of equal lengths: one lookup)
        const size_t nT = ixf.content.names.size();
        if (classOf.size() * nT > (size_t)4000000) deviceRank = false;
        else {
            vector<double> den(classOf.size() * nT);
            uint32_t ci = 0;
            for (auto &kv : classOf) {
                kv.second = ci;
                for (size_t t = 0; t < nT; ++t)
                    den[(size_t)ci * nT + t] = 1.0 + log2(ixf.freq[t] * double(uint32_t(kv.first - (p.protein ? p.K : p.K * 3) + 1)));   // Compare.hpp:1506-1511
                ++ci;
            }
            vector<uint32_t> rclass(nr);
            { uint32_t last = ~0u, cls = 0; for (uint64_t r = 0; r < nr; ++r) { if (b.rs.lengths[r] != last) { last = b.rs.lengths[r]; cls = classOf[last]; } rclass[r] = cls; } }
            uint64_t nEntries = 0;
            { ScopedTimerMt tm(g_ht.rank, g_ht.mu); if (kasa_batch_rank(ctx, den.data(), (uint32_t)classOf.size(), rclass.data(), p.threshold, (uint32_t)std::max(0, p.beasts), &nEntries, &nFlagged)) throwLast(); }
            if (nFlagged == 0 && !p.hostText) {
                // The text is written on the device (kasa_batch_text): neither the hits nor the rows cross PCIe, the buffer
                // that comes back is the file's next piece.  The host adds what only it knows: the specifiers and, per read
                // length, the perfect score.
                vector<float> best(classOf.size());
                for (auto &kv : classOf) best[kv.second] = bestScore(kv.first, p);
                kasa_text_params tp{};
                tp.format = p.fmt == Params::Tsv ? KASA_TEXT_TSV : p.fmt == Params::Json ? KASA_TEXT_JSON : p.fmt == Params::JsonL ? KASA_TEXT_JSONL : KASA_TEXT_KRAKEN;
                tp.beasts = (uint32_t)std::max(0, p.beasts); tp.firstRead = b.firstRead;
                tp.readNames = b.rs.nameBlob.data(); tp.readNameOff = b.rs.nameOff.data(); tp.readLen = b.rs.lengths.data();
                tp.bestScore = best.data(); tp.nClasses = (uint32_t)best.size();
                tp.coherence = p.coherence ? 1 : 0; tp.errorThreshold = (double)p.errorThreshold; tp.coherenceThreshold = p.coherenceThreshold;
                uint64_t nBytes = 0;
                { ScopedTimerMt tm(g_ht.text, g_ht.mu); if (kasa_batch_text(ctx, &tp, &nBytes)) throwLast(); }
                wb.flags.resize(nr);
                if (p.filter && kasa_batch_text_fetch(ctx, nullptr, nullptr, wb.flags.data())) throwLast();
                const size_t nPieces = out.fd < 0 ? 0 : (size_t)((nBytes + wb.pieceBytes - 1) / wb.pieceBytes);   // (--filter without -q: only the flags)
                out.begin(b.id, nPieces);
                b.flaggedByDevice = 0;
                for (size_t i = 0; i < nPieces; ++i) {
                    const unsigned slot = wb.turn++ % WorkerBuffers::TEXT_SLOTS;
                    const auto tTxt = std::chrono::steady_clock::now();
                    if (wb.written[slot].valid()) wb.written[slot].get();     // (the piece that was here is in the file)
                    tText += secondsSince(tTxt);
                    const uint64_t at = (uint64_t)i * wb.pieceBytes, len = std::min<uint64_t>(wb.pieceBytes, nBytes - at);
                    if (wb.text[slot].capacity < wb.pieceBytes) wb.text[slot].resize(wb.pieceBytes);
                    { ScopedTimerMt tm(g_ht.fetch, g_ht.mu); if (kasa_batch_text_fetch_range(ctx, wb.text[slot].data(), at, len)) throwLast(); }
                    wb.written[slot] = out.view(b.id, i, wb.text[slot].data(), (size_t)len);
                }
                if (p.filter) for (uint64_t r = 0; r < nr; ++r) if (wb.flags.data()[r]) b.flagged.push_back(b.firstRead + r);
                tDevice += secondsSince(tDev);
                return;
            }
            meta.resize(nr * 4); hits.resize(nEntries);
            if (kasa_batch_rank_fetch(ctx, meta.data(), hits.data())) throwLast();
        }
    }
    PcieBuf<uint64_t> &ro = wb.ro; PcieBuf<uint32_t> &tx = wb.tx; PcieBuf<float> &sc = wb.sc;
    if (!deviceRank || nFlagged) {
        uint64_t nnz = 0;
        if (kasa_batch_scores_size(ctx, &nnz)) throwLast();
        ro.resize(nr + 1); tx.resize(nnz); sc.resize(nnz);
        if (kasa_batch_scores_fetch(ctx, ro.data(), tx.data(), sc.data())) throwLast();
    }
    b.flaggedByDevice = nFlagged;
    // Compare::saveResults (Compare.hpp:2324-2443): what is waiting of an unfinished read joins read 0 when this batch ends
    // with a finished read; otherwise it stays and takes the unfinished last read's row as well -- the reference's own rule
    vector<uint32_t> mergedTax; vector<float> mergedSc; bool read0Merged = false;
    if (split) {
        if (!b.carry) throw std::runtime_error("a batch with an unfinished read has nowhere to leave its scores");
        vector<std::pair<uint32_t, float>> saved;
        if (b.head) saved = b.carry->take(b.id - 1);
        if (!saved.empty() && !b.tail) {
            const auto m = SplitCarry::merge(saved, tx.data() + ro[0], sc.data() + ro[0], ro[1] - ro[0]);
            for (const auto &e : m) { mergedTax.push_back(e.first); mergedSc.push_back(e.second); }
            read0Merged = true;
            saved.clear();
        }
        if (b.tail) {
            if (ro[nr] > ro[nr - 1]) saved = SplitCarry::merge(saved, tx.data() + ro[nr - 1], sc.data() + ro[nr - 1], ro[nr] - ro[nr - 1]);
            b.carry->put(b.id, std::move(saved));
        }
    }
    const uint64_t nPrinted = b.tail ? nr - 1 : nr;            // the unfinished read is printed by the batch that finishes it
    tDevice += secondsSince(tDev);
    const auto tTxt = std::chrono::steady_clock::now();
    const uint64_t slab = 1u << 15;
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(p.threads, (nPrinted + slab - 1) / slab));
    const uint64_t nSlabs = (nPrinted + slab - 1) / slab;
    out.begin(b.id, (size_t)nSlabs);
    vector<vector<uint64_t>> flagged(nSlabs);
    vector<std::exception_ptr> err(nt);
    std::atomic<uint64_t> nextSlab{0};
    auto work = [&](unsigned t) {
        try {
            Writer w(p, ixf.content, ixf.freq);
            for (;;) {
                const uint64_t sidx = nextSlab.fetch_add(1);
                if (sidx >= nSlabs) break;
                const uint64_t a = sidx * slab, e = std::min<uint64_t>(nPrinted, a + slab);
                string text;
                text.reserve((size_t)(e - a) * (p.fmt == Params::Json ? 900 : 600));
                for (uint64_t r = a; r < e; ++r) {
                    if (p.coherence) w.coherence = coherence[r];
                    if (r == 0 && read0Merged)
                        w.read(text, b.firstRead, b.rs.name(0), b.rs.lengths[0], mergedTax.data(), mergedSc.data(), mergedTax.size());
                    else if (deviceRank && !(meta[4 * r + 1] >> 31)) {
                        float maxV; std::memcpy(&maxV, &meta[4 * r + 2], 4);
                        w.readRanked(text, b.firstRead + r, b.rs.name(r), b.rs.lengths[r], hits.data() + meta[4 * r], meta[4 * r + 1], maxV);
                    } else
                        w.read(text, b.firstRead + r, b.rs.name(r), b.rs.lengths[r], tx.data() + ro[r], sc.data() + ro[r], ro[r + 1] - ro[r]);
                    if (p.filter && w.lastContaminated) flagged[sidx].push_back(b.firstRead + r);
                }
                out.submit(b.id, (size_t)sidx, std::move(text));
            }
        } catch (...) { err[t] = std::current_exception(); }
    };
    if (nt == 1) work(0);
    else { vector<std::thread> pool; for (unsigned t = 0; t < nt; ++t) pool.emplace_back(work, t); for (auto &th : pool) th.join(); }
    for (auto &e : err) if (e) std::rethrow_exception(e);
    for (auto &f : flagged) b.flagged.insert(b.flagged.end(), f.begin(), f.end());
    tText += secondsSince(tTxt);
}

// One input file: batches are formed by this thread while the device workers (one per device, each with its own context
// over the device's shared index) process the previous ones; per-read text leaves in batch order.  The profile tables of
// the devices are summed with one RCCL all-reduce (kasa_profile_allreduce) before the CSV is written.
static void identifyFile(Params p, const IndexFiles &ixf, const vector<int> &devSlots, const string &input, const string &input2,
                         const string &rtt, const string &profile)
{
    p.input = input; p.input2 = input2; p.rtt = rtt; p.profile = profile;
    const auto tStart = std::chrono::steady_clock::now();
    g_t0 = tStart; mark("file begins");
    const bool wantRows = !p.rtt.empty() || p.filter;
    const size_t nDev = devSlots.size();
    vector<kasa_ctx *> ctx(nDev, nullptr);
    struct CtxGuard { vector<kasa_ctx *> &c; ~CtxGuard() { for (auto *x : c) kasa_ctx_destroy(x); } } guard{ctx};
    for (size_t d = 0; d < nDev; ++d)
        if (kasa_ctx_create(ixf.onDevice[(size_t)devSlots[d]], p.kHigh, p.kLow, p.frames, ixf.lut.empty() ? nullptr : ixf.lut.data(), &ctx[d])) throwLast();
    vector<vector<kasa_ctx *>> partCtx(nDev);                          // a partitioned index: one context per partition and device
    struct PartGuard { vector<vector<kasa_ctx *>> &c; ~PartGuard() { for (auto &v : c) for (auto *x : v) kasa_ctx_destroy(x); } } partGuard{partCtx};
    if (!ixf.parts.empty())
        for (size_t d = 0; d < nDev; ++d)
            for (kasa_index *ix : ixf.parts[(size_t)devSlots[d]]) {
                kasa_ctx *c = nullptr;
                if (kasa_ctx_create(ix, p.kHigh, p.kLow, p.frames, ixf.lut.empty() ? nullptr : ixf.lut.data(), &c)) throwLast();
                partCtx[d].push_back(c);
            }
    // device batches: up to 2^32 k-mers, and what fits the HBM that is free next to the index
    uint64_t maxKmersPerBatch = 3000000000ull;
    {
        uint64_t freeB = 0, totalB = 0;
        if (kasa_device_memory(p.devices[(size_t)devSlots[0]], &freeB, &totalB)) throwLast();
        const uint64_t per = kasa_batch_bytes_per_query(ctx[0]);
        if (per) maxKmersPerBatch = std::max<uint64_t>(1u << 20, std::min<uint64_t>(maxKmersPerBatch, (uint64_t)(0.8 * (double)freeB) / per));
        if (const char *e = getenv("KASA_MAX_BATCH_KMERS")) maxKmersPerBatch = std::max<uint64_t>(1, (uint64_t)atoll(e));   // tests: force several batches
    }
    mark("contexts made");
    OrderedOut out;
    vector<WorkerBuffers> wbs(nDev);                                   // (after `out`: their destructors wait for its writer)
    // the page-locked text buffers are made while the first chunk is parsed
    std::thread textPrep;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joinTextPrep{textPrep};
    if (!p.rtt.empty() && !p.hostRank && !p.hostText) textPrep = std::thread([&wbs, &p] { try { for (size_t i = 0; i < wbs.size(); ++i) { (void)kasa_thread_device(p.devices[i]); wbs[i].prepareText(); } } catch (...) {} });   // (a fresh thread stands on device 0: the buffers are page-locked for the worker's device)   // (what is missing is made, or fails, when a worker needs it)
    Batcher batcher(p, ixf, wantRows, maxKmersPerBatch);
    mark("first chunk parsed");
    p.protein = batcher.protein;
    for (auto *c : ctx) if (kasa_ctx_set_protein(c, p.protein ? 1 : 0)) throwLast();
    for (auto &v : partCtx) for (auto *c : v) if (kasa_ctx_set_protein(c, p.protein ? 1 : 0)) throwLast();
    if (wantRows && !p.hostRank && !p.hostText) {                     // what the device prints for a taxon (kasa_batch_text)
        vector<uint64_t> off(ixf.content.names.size() + 1, 0);
        string blob;
        for (size_t t = 0; t < ixf.content.names.size(); ++t) { blob += ixf.content.names[t]; off[t + 1] = blob.size(); }
        for (auto *c : ctx) if (kasa_ctx_set_taxa_text(c, ixf.content.taxids.data(), blob.data(), off.data())) throwLast();
    }
    // the device buffers of the first batch are allocated while the input is parsed (hipMalloc: 25-90 ms per GB here)
    std::thread reserver;
    Joiner joinReserver{reserver};
    {
        uint64_t estQ = 0, estB = 0;
        batcher.estimateFirstBatch(estQ, estB);
        if (estQ > 0 && !getenv("KASA_NO_RESERVE"))
            reserver = std::thread([&ctx, estQ, estB, wantRows] { for (auto *c : ctx) (void)kasa_ctx_reserve(c, estQ, estB, wantRows ? 1 : 0); });   // (a failure shows when the batch allocates)
    }
    if (!p.rtt.empty()) {
        out.fd = ::open(p.rtt.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (out.fd < 0) throw std::runtime_error("Readwise output file could not be created!");
        if (p.fmt == Params::Tsv) out.put(p.coherence ? "#Read number\tSpecifier from input file\tMatched taxa\tNames\tScores{relative,k-mer}\tError\tCoherence\n"
                                                      : "#Read number\tSpecifier from input file\tMatched taxa\tNames\tScores{relative,k-mer}\tError\n");
        else if (p.fmt == Params::Json) out.put("[\n");
    }
    if (!p.profile.empty() && !std::ofstream(p.profile)) throw std::runtime_error("Profile file couldn't be opened for writing!");

    // queue of formed batches (bounded: one waiting per device) and results that leave in batch order
    std::mutex mu;
    std::condition_variable cvWork, cvSpace, cvDone;
    std::deque<std::unique_ptr<Batch>> todo;
    std::map<uint64_t, std::unique_ptr<Batch>> finished;
    bool noMore = false;
    std::exception_ptr failure;
    vector<double> tDevice(nDev, 0.0), tText(nDev, 0.0);
    std::atomic<uint64_t> totalKmers{0};
    SplitCarry carry;
    auto worker = [&](size_t d) {
        WorkerBuffers &wb = wbs[d];
        try {
            for (;;) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cvWork.wait(lk, [&] { return !todo.empty() || noMore || failure; });
                    if (failure || todo.empty()) return;
                    b = std::move(todo.front()); todo.pop_front();
                    cvSpace.notify_all();
                }
                mark("device takes batch", b->id);
                runBatch(p, ixf, ctx[d], partCtx[d], *b, wantRows, tDevice[d], tText[d], wb, out);
                mark("device done with batch", b->id);
                totalKmers += b->kmers;
                b->rs = ReadSet();                                   // the reads are done with
                std::lock_guard<std::mutex> lk(mu);
                finished[b->id] = std::move(b);
                cvDone.notify_all();
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (!failure) failure = std::current_exception();       // like Compare.hpp:1060-1067: parked, rethrown by the driver thread
            out.abandon(); carry.abandon();
            cvDone.notify_all(); cvSpace.notify_all(); cvWork.notify_all();
        }
    };
    vector<std::thread> pool;
    for (size_t d = 0; d < nDev; ++d) pool.emplace_back(worker, d);
    vector<uint64_t> contaminants;              // --filter: read numbers, ascending
    uint64_t nBatches = 0, written = 0;
    auto drain = [&](bool all) {                 // write finished batches in order (caller holds no lock)
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            auto it = finished.find(written);
            if (it == finished.end()) {
                if (!all || written == nBatches || failure) return;
                cvDone.wait(lk, [&] { return finished.count(written) || failure; });
                continue;
            }
            std::unique_ptr<Batch> b = std::move(it->second);
            finished.erase(it);
            lk.unlock();
            contaminants.insert(contaminants.end(), b->flagged.begin(), b->flagged.end());
            ++written;
            lk.lock();
        }
    };
    try {
        for (;;) {
            std::unique_ptr<Batch> b(new Batch());
            if (!batcher.next(*b)) break;
            b->carry = &carry;
            if (p.verbose) std::cout << "OUT: Batch of " << b->rs.size() << " reads" << (b->head ? ", the first goes on from the batch before" : "") << (b->tail ? ", the last goes on in the next" : "") << std::endl;
            ++nBatches;
            mark("batch formed", b->id);
            if (reserver.joinable()) { reserver.join(); mark("device buffers reserved"); }
            if (textPrep.joinable()) { textPrep.join(); mark("text buffers page-locked"); }
            {
                std::unique_lock<std::mutex> lk(mu);
                cvSpace.wait(lk, [&] { return todo.size() < nDev || failure; });
                if (failure) break;
                todo.push_back(std::move(b));
                cvWork.notify_one();
            }
            drain(false);
        }
    } catch (...) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failure) failure = std::current_exception();
    }
    { std::lock_guard<std::mutex> lk(mu); noMore = true; cvWork.notify_all(); }
    drain(true);
    for (auto &t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
    const uint64_t nReads = batcher.nextRead;
    mark("workers joined");
    if (!p.rtt.empty()) { out.drainWriter(); out.stopWriter(); mark("writer drained"); if (p.fmt == Params::Json) out.put("\n]"); ::close(out.fd); out.fd = -1; }
    if (p.filter) filterReads(p, contaminants);
    // profile: one RCCL all-reduce over the devices' tables, then device 0's copy
    if (nDev > 1 || getenv("KASA_FORCE_ALLREDUCE")) {        // (the variable: tests run the reduce with a single rank)
        vector<ncclComm_t> comms(nDev);
        vector<int> devs;
        for (size_t d = 0; d < nDev; ++d) devs.push_back(p.devices[(size_t)devSlots[d]]);
        if (ncclCommInitAll(comms.data(), (int)nDev, devs.data()) != ncclSuccess) throw std::runtime_error("RCCL communicator could not be created");
        vector<std::thread> red; vector<int> rcs(nDev, 0); vector<string> msgs(nDev);
        for (size_t d = 0; d < nDev; ++d) red.emplace_back([&, d] { rcs[d] = kasa_profile_allreduce(ctx[d], comms[d]); if (rcs[d]) msgs[d] = kasa_last_error(); });
        for (auto &t : red) t.join();
        for (auto &c : comms) ncclCommDestroy(c);
        for (size_t d = 0; d < nDev; ++d) if (rcs[d]) throw std::runtime_error(msgs[d]);
    }
    const int nK = p.kHigh - p.kLow + 1;
    vector<double> all((size_t)nK * ixf.content.names.size()); vector<uint64_t> uniq(all.size()), tot(all.size());
    if (kasa_profile_fetch(ctx[0], all.data(), uniq.data(), tot.data())) throwLast();
    if (!p.profile.empty()) writeProfile(p.profile, p, ixf.content, all, uniq, tot, ixf.freqAll, totalKmers.load(), nReads);
    mark("profile written");
    if (p.verbose && g_ht.on)
        std::cout << "OUT: host timing: read " << g_ht.read << " s, cut " << g_ht.cut << " s, parse " << g_ht.parse << " s, merge " << g_ht.merge
                  << " s, batch forming " << g_ht.form << " s, output write " << g_ht.write << " s; upload " << g_ht.upload << " s, device " << g_ht.compute
                  << " s (encode " << g_ht.encode << ", sort " << g_ht.sort << ", lookup + score " << g_ht.score << "), ranking " << g_ht.rank << " s, text " << g_ht.text << " s, text fetch " << g_ht.fetch << " s" << std::endl;
    if (p.verbose && g_ht.on) {
        std::cout << "OUT: device stages (HIP events, ms):";
        const char *nm[] = {"encode", "sort", "lookup", "group", "regroup", "score"};
        for (int st = 0; st < 6; ++st) { double ms = 0; uint64_t n = 0; if (!kasa_ctx_stage_ms(ctx[0], st, &ms, &n)) std::cout << " " << nm[st] << " " << ms; }
        std::cout << std::endl;
    }
    if (p.verbose) {
        double ident = 0; for (size_t t = 1; t < ixf.content.names.size(); ++t) ident += all[(size_t)(nK - 1) * ixf.content.names.size() + t];
        double dev = 0, txt = 0; for (size_t d = 0; d < nDev; ++d) { dev = std::max(dev, tDevice[d]); txt = std::max(txt, tText[d]); }
        std::cout << "OUT: Number of k-mers in input: " << totalKmers.load() << " of which " << ident / (double)totalKmers.load() * 100. << " % were identified." << std::endl;
        std::cout << "OUT: Time fastq: " << batcher.parseSeconds << " s (" << p.threads << " threads)\nOUT: Time compare: " << dev << " s (" << nDev << " device" << (nDev > 1 ? "s" : "")
                  << ")\nOUT: Time output: " << txt << " s\nOUT: Time file: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count() << " s" << std::endl;   // Compare.hpp:3689-3690
        // (ours) how busy the device was: the five stages' HIP-event time of device 0 against the file's wall time.  The
        // reference's -m cuts batches for ITS memory budget; here it only decides how many reads meet the device at once.
        double stagesMs = 0;
        for (int st = 0; st < 6; ++st) { double ms = 0; uint64_t n = 0; if (!kasa_ctx_stage_ms(ctx[0], st, &ms, &n)) stagesMs += ms; }
        const double fileS = std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count();
        if (stagesMs > 0 && fileS > 0) {
            const double util = stagesMs / 1000. / fileS * 100.;
            std::cout << "OUT: Device: " << nBatches << " batch" << (nBatches == 1 ? "" : "es") << " of " << (nBatches ? nReads / nBatches : 0) << " reads on average; its identify stages ran "
                      << stagesMs / 1000. << " s = " << util << " % of the file's time";
            if (util < 50. && nBatches > 1 && nReads / nBatches < 2000000)
                std::cout << " -- batches this small leave it waiting for the host: a larger -m (the results do not depend on it unless they are compared with kASA's byte for byte) keeps it busier";
            std::cout << std::endl;
        }
    }
}

// Utilities::gatherFilesFromPath (Utilities.hpp:154-264): a path ending in '/' names every file of that folder
static vector<string> gatherFiles(const string &path)
{
    vector<string> files;
    if (!path.empty() && path.back() == '/') {
        if (DIR *d = opendir(path.c_str())) {
            while (dirent *e = readdir(d)) { const string n(e->d_name); if (n != "." && n != "..") files.push_back(path + n); }
            closedir(d);
        }
        std::sort(files.begin(), files.end());
    } else files.push_back(path);
    return files;
}

static const char *formatEnding(Params::Fmt f)    // Compare.hpp:367-381
{
    switch (f) { case Params::Kraken: return ".ktsv"; case Params::Json: return ".json"; case Params::JsonL: return ".jsonl"; default: return ".tsv"; }
}

// ---------------------------------------------------------------------------------------------------
// main
// ---------------------------------------------------------------------------------------------------
// `kASA --parameters <file>`: the reference reads its whole command line from a YAML-like file instead (source/main.cpp:264-302,
// source/utils/Utilities.hpp:1114-1400: `Key: value` per line, '#' comments, quotes dropped, booleans "true").  Here the keys that
// concern `identify` become the flags they stand for and pass through the one parser below; keys of other modes are ignored
// as the reference ignores them in this mode.
static vector<string> argsFromYaml(const string &exe, const string &file)
{
    std::ifstream in(file);
    if (!in) throw std::runtime_error("Config file not found!");                    // Utilities.hpp:1115-1117
    auto strip = [](string v) {
        string o; for (char c : v) if (c != '"') o += c;
        size_t b = 0; while (b < o.size() && (o[b] == ' ' || o[b] == '\t')) ++b;
        size_t e = o.size(); while (e > b && (o[e - 1] == ' ' || o[e - 1] == '\t' || o[e - 1] == '\r' || o[e - 1] == '\n')) --e;
        return o.substr(b, e - b);
    };
    string mode = "identify", kH, kL, alphaFile, alphaIdx, line;
    vector<string> rest;
    auto flag = [&](const string &f, const string &v) { if (v == "true") rest.push_back(f); };
    auto opt = [&](const string &f, const string &v) { if (!v.empty()) { rest.push_back(f); rest.push_back(v); } };
    while (std::getline(in, line)) {
        size_t b = 0; while (b < line.size() && (line[b] == ' ' || line[b] == '\t')) ++b;
        if (b >= line.size() || line[b] == '#') continue;
        const size_t colon = line.find(':', b);
        if (colon == string::npos) continue;
        const string key = strip(line.substr(b, colon - b)), val = strip(line.substr(colon + 1));
        if (key == "Mode") mode = val;
        else if (key == "Index") opt("-d", val);
        else if (key == "ContentFile") opt("-c", val);
        else if (key == "kHigh") kH = val;
        else if (key == "kLow") kL = val;
        else if (key == "NumberOfThreads") opt("-n", val == "-1" ? std::to_string(std::max(1u, std::thread::hardware_concurrency())) : val);
        else if (key == "AvailableRAMinGB") opt("-m", val);
        else if (key == "FilePathForTemporaryFiles") opt("-t", val);
        else if (key == "CallIndex") opt("-x", val);
        else if (key == "Verbose") flag("-v", val);
        else if (key == "AlphabetFile") alphaFile = val;
        else if (key == "AlphabetIndex") alphaIdx = val;
        else if (key == "InputFileOrFolder") opt("-i", val);
        else if (key == "PairedEnd-First") opt("-1", val);
        else if (key == "PairedEnd-Second") opt("-2", val);
        else if (key == "AlreadyTranslated") flag("-z", val);
        else if (key == "One") flag("--one", val);
        else if (key == "Three") flag("--three", val);
        else if (key == "Six") flag("--six", val);
        else if (key == "ProfileOutputfile") opt("-p", val);
        else if (key == "ReadIDtoTaxIDOutputfile") opt("-q", val);
        else if (key == "ReadIDtoTaxIDOutputFormat") { if (val == "json" || val == "jsonl" || val == "kraken" || val == "tsv") rest.push_back("--" + val); }
        else if (key == "UseRAMOnly") flag("-r", val);
        else if (key == "NumberOfTaxaPerRead") opt("-b", val);
        else if (key == "UniqueKmersOnly") flag("-e", val);
        else if (key == "ThresholdForScore") opt("--threshold", val);
        else if (key == "PrintCoverage") flag("--coverage", val);
        else if (key == "Filter") {                                                  // "clean contaminated"; "_ _" = none (Utilities.hpp:1326-1334)
            std::stringstream ss(val); string f0, f1; ss >> f0 >> f1;
            if (!f1.empty() && (f0 != "_" || f1 != "_")) { rest.push_back("--filter"); rest.push_back(f0); rest.push_back(f1); }
        }
        else if (key == "ErrorThreshold") opt("--errorThreshold", val);
        else if (key == "Gzip") flag("--gzip", val);
    }
    vector<string> out = {exe, mode};
    if (!kH.empty() && !kL.empty()) { out.push_back("-k"); out.push_back(kH); out.push_back(kL); }
    else if (!kH.empty()) { out.push_back("--kH"); out.push_back(kH); }
    else if (!kL.empty()) { out.push_back("--kL"); out.push_back(kL); }
    if (!alphaFile.empty()) { out.push_back("-a"); out.push_back(alphaFile); out.push_back(alphaIdx.empty() ? "1" : alphaIdx); }
    out.insert(out.end(), rest.begin(), rest.end());
    return out;
}

static int run(int argc, char **argv)
{
    vector<string> a(argv, argv + argc);
    if (argc >= 3 && a[1] == "--parameters") {                   // source/main.cpp:264
        a = argsFromYaml(a[0], a[2]);
        argc = (int)a.size();
    }
    std::cout << "OUT: kasa_identify (MI355X path of kASA identify)\nOUT: ";
    for (auto &s : a) std::cout << s << " ";
    std::cout << std::endl;
    if (argc >= 4 && a[1] == "parse-dump") {                     // test tap: what the parsers make of a file (no device involved)
        const unsigned nt = (unsigned)std::stoul(a[3]);
        auto dump = [](const ReadSet &rs, size_t from) {
            for (size_t r = from; r < rs.size(); ++r) {
                std::cout << rs.name(r) << "\t" << rs.lengths[r] << "\t";
                std::cout.write((const char *)rs.bases.data() + rs.off[r], rs.off[r + 1] - rs.off[r]);
                std::cout << "\n";
            }
        };
        const bool quiet = argc >= 5 && a[4] == "quiet";         // timing only
        if (!quiet) {
            std::cout << "== whole file\n";
            ReadSet rs = readInput(a[2], false, nt); std::cout << "protein=" << rs.protein << "\n"; dump(rs, 0);
            std::cout << "== streamed\n";
        }
        const auto t0 = std::chrono::steady_clock::now();
        ChunkReader cr(a[2]);
        const char *chunk; size_t chunkBytes; ReadSet all; vector<ReadSet> parts;
        while (cr.next(chunk, chunkBytes, false)) parsePiece(chunk, chunkBytes, cr.fasta, nt, 1u << 20, all, parts, cr.chunkStart);
        std::cout << "protein=" << cr.protein << " fasta=" << cr.fasta << "\n";
        if (!quiet) dump(all, 0);
        else std::cout << all.size() << " reads, " << all.bases.size() << " bases in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s; read " << g_ht.read
                       << " cut " << g_ht.cut << " parse " << g_ht.parse << " merge " << g_ht.merge << "\n";
        return 0;
    }
    if (argc >= 6 && a[1] == "pieces-dump") {                    // test tap: the pieces of every long record -- <file> <threads> <frames> <K> (no device involved)
        const unsigned nt = (unsigned)std::stoul(a[3]);
        ChunkReader cr(a[2]);
        const char *chunk; size_t chunkBytes; ReadSet all; vector<ReadSet> parts;
        while (cr.next(chunk, chunkBytes, false)) parsePiece(chunk, chunkBytes, cr.fasta, nt, 1u << 20, all, parts, cr.chunkStart);
        std::sort(all.longRecs.begin(), all.longRecs.end(), [](const ReadSet::LongRec &x, const ReadSet::LongRec &y) { return x.read < y.read; });
        for (const ReadSet::LongRec &lr : all.longRecs) {
            const Pieces pc = readerPieces(lr, cr.fasta, cr.protein, std::stoi(a[4]), std::stoi(a[5]), false);
            std::cout << lr.read << "\t";
            for (size_t i = 0; i < pc.cut.size(); ++i) std::cout << (i ? "," : "") << pc.cut[i];
            std::cout << "\t";
            for (size_t i = 0; i < pc.add.size(); ++i) std::cout << (i ? "," : "") << pc.add[i];
            std::cout << "\n";
        }
        return 0;
    }
    if (argc < 2 || (a[1] != "identify" && a[1] != "identify_multiple")) throw std::runtime_error("only the modes `identify` and `identify_multiple` are available on this path");
    Params p;
    p.mode = a[1];
    int frameFlags = 0;
    for (int i = 2; i < argc; ++i) {
        const string &s = a[i];
        auto next = [&]() -> string { if (i + 1 >= argc) throw std::runtime_error("missing value after " + s); return a[++i]; };
        if (s == "-c" || s == "--content") p.content = next();
        else if (s == "-d" || s == "--database") p.index = next();
        else if (s == "-i" || s == "--input" || s == "-1") { p.input = next(); if (!std::ifstream(p.input) && p.input.back() != '/') throw std::runtime_error("Input file not found"); }
        else if (s == "-2") { p.input2 = next(); if (!std::ifstream(p.input2)) throw std::runtime_error("Input file not found"); }
        else if (s == "-q" || s == "--rtt") p.rtt = next();
        else if (s == "-p" || s == "--profile") p.profile = next();
        else if (s == "-k") { p.kSetByUser = true; p.kHigh = std::stoi(next()); p.kLow = std::stoi(next()); if (p.kHigh > 25) p.kHigh = 25; if (p.kLow < 1) p.kLow = 1; if (p.kLow > p.kHigh) std::swap(p.kLow, p.kHigh); }
        else if (s == "--kH") { p.kSetByUser = true; p.kHigh = std::min(25, std::stoi(next())); }
        else if (s == "--kL") { p.kLow = std::max(1, std::stoi(next())); }
        else if (s == "-b" || s == "--beasts") p.beasts = std::stoi(next());
        else if (s == "--json") p.fmt = Params::Json;
        else if (s == "--jsonl") p.fmt = Params::JsonL;
        else if (s == "--tsv") p.fmt = Params::Tsv;
        else if (s == "--kraken") p.fmt = Params::Kraken;
        else if (s == "--threshold") p.threshold = std::stof(next());
        else if (s == "--six") { p.frames = 6; ++frameFlags; }
        else if (s == "--three") { p.frames = 3; ++frameFlags; }
        else if (s == "--one") { p.frames = 1; ++frameFlags; }
        else if (s == "-e" || s == "--unique") p.unique = true;
        else if (s == "--coverage") p.coverage = true;
        else if (s == "-v" || s == "--verbose") p.verbose = true;
        else if (s == "--device") p.devices = {std::stoi(next())};
        else if (s == "--devices") {                                          // read shards over several GPUs, index replicated
            p.devices.clear();
            std::stringstream ss(next()); string tok;
            while (std::getline(ss, tok, ',')) if (!tok.empty()) p.devices.push_back(std::stoi(tok));
            if (p.devices.empty()) throw std::runtime_error("--devices needs a list like 0,1,2,3");
        }
        else if (s == "-r" || s == "--ram") p.ram = true;                    // the index always lives in HBM; -r only enters the batch budget
        else if (s == "-n" || s == "--threads") { p.threads = (unsigned)std::max(1, std::stoi(next())); p.refThreads = (int)p.threads; }
        else if (s == "-m" || s == "--memory") { const string v = next(); p.memoryGiB = v == "inf" ? (1 << 30) : std::stoi(v); }   // main.cpp:438-447
        else if (s == "-t" || s == "--temp" || s == "-x" || s == "--callidx") next();
        else if (s == "--host-rank") p.hostRank = true;
        else if (s == "--host-text") p.hostText = true;
        else if (s == "--allow-device-split") p.allowDeviceSplit = true;
        else if (s == "--filter") { p.filter = true; p.filterClean = next(); p.filterCont = next(); }
        else if (s == "--errorThreshold") p.errorThreshold = std::stof(next());
        else if (s == "--gzip") p.gzipOut = true;                                                  // main.cpp:570-572
        else if (s == "-a" || s == "--alphabet") { p.codonFile = next(); p.codonId = next(); }
        else if (s == "--coherence") p.coherence = true;                                        // main.cpp:576-581
        else if (s == "--coherenceThreshold") p.coherenceThreshold = std::stof(next());
        else if (s == "--visualize" || s == "-z")
            throw std::runtime_error("parameter " + s + " is not supported by the MI355X identify path");
        else throw std::runtime_error("Some unknown parameter has been inserted, please check your command line.");
    }
    if (frameFlags >= 2) throw std::runtime_error("You'll have to decide between using one, three, or six frames. Currently, more than one option was chosen. Please check your parameters!"); // main.cpp:618-620
    std::ifstream info(p.index + "_info.txt");
    if (!info) throw std::runtime_error("Info file for this index can not be found!");
    IndexFiles ixf;
    uint64_t vecType = 0; info >> ixf.nRec; info >> vecType;
    ixf.recBytes = vecType == 128 ? 20 : (vecType == 3 ? 6 : 12);   // 128: k <= 25 index, 3: halved index of shrink strategy 2
    if (vecType == 128) {
        p.K = 25;
        if (!p.kSetByUser) p.kHigh = 25;                               // main.cpp:1065-1067
    } else {
        if (p.kHigh > 12) { std::cerr << "WARNING: This index can not be used with a k higher than 12! Setting to this maximum..." << std::endl; p.kHigh = 12; }
        if (p.kLow > 12) p.kLow = 12;
    }
    if (p.content.empty()) p.content = p.index + "_content.txt";
    ixf.content = loadContent(p.content);
    ixf.freqAll = loadFreq(p.index, ixf.content.names.size(), p.kHigh, p.kLow);
    ixf.freq.assign(ixf.freqAll.begin(), ixf.freqAll.begin() + (std::ptrdiff_t)ixf.content.names.size());   // level 0: k = kHigh
    for (size_t t = 1; t < ixf.content.names.size(); ++t) ixf.nameBytes += ixf.content.names[t].size();
    if (!p.codonFile.empty()) ixf.lut = codonTableFromFile(p.codonFile, p.codonId);

    // index + trie files as they are on disk, once per device (the index object is immutable and shared by every context there)
    const int fd = open(p.index.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("The index file cannot be found!");
    void *rec = mmap(nullptr, ixf.nRec * ixf.recBytes, PROT_READ, MAP_PRIVATE, fd, 0);
    if (rec == MAP_FAILED) throw std::runtime_error("The index file cannot be mapped!");
    {
        std::ifstream ts(p.index + "_trie.txt"); std::ifstream tf(p.index + "_trie", std::ios::binary);
        if (!ts || !tf) throw std::runtime_error("The trie file cannot be found!");
        uint64_t m = 0; ts >> m; ixf.tp.resize(m); ixf.tc.resize(m);
        vector<char> raw(m * 12); tf.read(raw.data(), (std::streamsize)raw.size());
        for (uint64_t i = 0; i < m; ++i) { memcpy(&ixf.tc[i], &raw[i * 12], 8); memcpy(&ixf.tp[i], &raw[i * 12 + 8], 4); }
    }
    uint64_t maxPart = 0xFFFFFFF0ull - 1;                               // what one index object holds (kasa_index_create)
    if (const char *e = getenv("KASA_INDEX_PART_RECORDS")) maxPart = std::max<uint64_t>(1, (uint64_t)atoll(e));   // tests: partitions of a small index
    if (ixf.nRec <= maxPart) {
        for (int dev : p.devices) {
            kasa_index *ix = nullptr;
            if (kasa_index_create(dev, rec, ixf.nRec, ixf.recBytes, ixf.tp.data(), ixf.tc.data(), ixf.tp.size(), ixf.content.taxids.data(),
                                  (uint32_t)ixf.content.taxids.size(), &ix)) throwLast();
            ixf.onDevice.push_back(ix);
        }
    } else {
        // about equal partitions, as few as hold the index, cut between `_trie` entries
        const uint64_t m = ixf.tp.size();
        uint64_t nParts = (ixf.nRec + maxPart - 1) / maxPart;
        vector<uint64_t> tLo, rLo;                                         // first trie entry and first record of every partition, + the ends
        for (;; ++nParts) {
            tLo.assign(1, 0); rLo.assign(1, 0);
            uint64_t t = 0, r = 0; bool ok = true;
            for (uint64_t j = 0; j < nParts; ++j) {
                const uint64_t r0 = r;
                if (j + 1 == nParts) while (t < m) r += ixf.tc[t++];
                else { const uint64_t target = ixf.nRec / nParts * (j + 1); while (t < m && (r == r0 || r + ixf.tc[t] <= target)) r += ixf.tc[t++]; }
                if (r == r0 || r - r0 > maxPart) ok = false;
                tLo.push_back(t); rLo.push_back(r);
            }
            if (ok) { if (r != ixf.nRec) throw std::runtime_error("the trie file does not add up to the index's records"); break; }
            if (nParts >= m) throw std::runtime_error("the index cannot be cut into partitions of at most " + std::to_string(maxPart) + " records between its trie entries");
        }
        if (p.coherence) throw std::runtime_error("--coherence over an index of 2^32 records and more is not supported");
        ixf.parts.resize(p.devices.size());
        for (uint64_t j = 0; j < nParts; ++j) ixf.cuts.push_back(j == 0 ? 0 : (uint64_t)ixf.tp[tLo[j]]);
        for (size_t d = 0; d < p.devices.size(); ++d) {
            for (uint64_t j = 0; j < nParts; ++j) {
                kasa_index *ix = nullptr;
                if (kasa_index_create(p.devices[d], (const char *)rec + rLo[j] * (uint64_t)ixf.recBytes, rLo[j + 1] - rLo[j], ixf.recBytes, ixf.tp.data() + tLo[j], ixf.tc.data() + tLo[j],
                                      tLo[j + 1] - tLo[j], ixf.content.taxids.data(), (uint32_t)ixf.content.taxids.size(), &ix)) throwLast();
                ixf.parts[d].push_back(ix);
            }
            ixf.onDevice.push_back(ixf.parts[d][0]);
        }
        if (p.verbose) std::cout << "OUT: Index of " << ixf.nRec << " records in " << nParts << " partitions on every device" << std::endl;
    }
    munmap(rec, ixf.nRec * ixf.recBytes); close(fd);
    if (p.threads == 0) {
        // all CPUs this process may really use: the machine's, capped by a cgroup CPU quota (a container that sees 256 CPUs
        // and is throttled to 16 runs slower with 128 threads than with 16)
        unsigned cpus = std::max(1u, std::min(128u, std::thread::hardware_concurrency()));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[64] = {0}; double period = 0;
            if (fscanf(f, "%63s %lf", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) {
                const double q = atof(a) / period;
                if (q >= 1.0) cpus = std::min(cpus, (unsigned)(q + 0.5));
            }
            fclose(f);
        }
        p.threads = cpus;
    }

    vector<int> allSlots;
    for (size_t d = 0; d < p.devices.size(); ++d) allSlots.push_back((int)d);
    const vector<string> files = gatherFiles(p.input);
    if (files.empty()) throw std::runtime_error("Input file not found");
    auto outputsOf = [&](const string &file, string &rtt, string &prof) {   // main.cpp:1296-1313, Compare.hpp:3052
        string name = file.substr(p.input.back() == '/' ? p.input.size() : 0);
        const size_t dot = name.rfind('.');
        if (dot != string::npos && dot > 0) name.resize(dot);
        rtt = p.rtt.empty() ? "" : p.rtt + name + formatEnding(p.fmt);
        prof = p.profile.empty() ? "" : p.profile + name + ".csv";
    };
    if (p.mode == "identify") {
        // one file (or every file of a folder, one after the other): every batch goes to the next free device
        for (const string &f : files) {
            string rtt = p.rtt, prof = p.profile;
            if (files.size() > 1) outputsOf(f, rtt, prof);
            identifyFile(p, ixf, allSlots, f, files.size() > 1 ? "" : p.input2, rtt, prof);
        }
    } else {
        // identify_multiple (main.cpp:1118-1334): the files are a job queue; every worker runs a whole file with a context of
        // its own over the shared index of its device (two workers per device keep it busy while the other one parses / writes)
        std::atomic<size_t> nextFile{0};
        std::mutex mu; std::exception_ptr failure;
        const size_t nWorkers = std::min(files.size(), p.devices.size() * 2);
        Params pw = p;
        pw.threads = std::max(1u, p.threads / (unsigned)nWorkers);
        vector<std::thread> pool;
        for (size_t w = 0; w < nWorkers; ++w)
            pool.emplace_back([&, w] {
                try {
                    for (;;) {
                        const size_t i = nextFile.fetch_add(1);
                        if (i >= files.size()) return;
                        string rtt, prof;
                        outputsOf(files[i], rtt, prof);
                        identifyFile(pw, ixf, {(int)(w % p.devices.size())}, files[i], "", rtt, prof);
                    }
                } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!failure) failure = std::current_exception(); }
            });
        for (auto &t : pool) t.join();
        if (failure) std::rethrow_exception(failure);
    }
    return 0;
}

int main(int argc, char **argv)
{
    try { return run(argc, argv); }
    catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; } // main.cpp:1717-1720
}

// kasa_text.h -- the per-read result file written ON THE DEVICE (second half): the bytes
// Compare::scoringFunc's printing loops put into the -q file (Compare.hpp:1526-1872) for reads that kasa_batch_rank has ranked,
// in the reference's number formats -- itostr (utils/iToStr.hpp:35-114) and the Grisu2 shortest-digits dtoa
// (utils/dToStr.h:427-456; Loitsch, PLDI 2010), both integer-only.  One lane per read, two passes of the same code: the
// first counts the bytes of every read, an exclusive scan turns the counts into offsets, the second writes -- so the buffer
// that crosses PCIe IS the file's next piece.
//
// Included by kasa_hip.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace kasa_text {

struct DiyFp { uint64_t f; int e; };
__device__ const DiyFp kPowers[87] = {
#include "../host/grisu_powers.inc"
};
__device__ const uint32_t kPow10[10] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u, 1000000000u};

enum { FMT_TSV = 0, FMT_JSON = 1, FMT_JSONL = 2, FMT_KRAKEN = 3 };

struct Entry { uint32_t tax; float score; double rel; };       // = RankEntry of kasa_batch_rank

struct Args {
    int fmt;
    uint32_t beasts;
    uint64_t firstRead;
    uint32_t nReads;
    const char *readNames; const uint64_t *readNameOff;       // specifier of read r = readNames[readNameOff[r] .. readNameOff[r+1])
    const uint32_t *readLen;                                   // "Length"
    const uint32_t *readClass; const float *best;              // best[readClass[r]] = the perfect score of a read of that length
    const uint4 *meta; const Entry *entries;                   // kasa_batch_rank's results
    const uint32_t *taxIds; const char *taxNames; const uint64_t *taxNameOff;
    int coherence; const float *cohScores;
    double errorThreshold; float coherenceThreshold;
};

// The sink of the formatting code: counts, or writes bytes at `p`.
template <bool WRITE> struct Sink {
    char *p; uint64_t n;
    __device__ __forceinline__ void ch(char c) { if (WRITE) p[n] = c; ++n; }
    template <int N> __device__ __forceinline__ void lit(const char (&s)[N])
    {
        if (WRITE) {
#pragma unroll
            for (int i = 0; i < N - 1; ++i) p[n + i] = s[i];
        }
        n += N - 1;
    }
    __device__ __forceinline__ void bytes(const char *s, uint64_t len)
    {
        if (WRITE) for (uint64_t i = 0; i < len; ++i) p[n + i] = s[i];
        n += len;
    }
    __device__ __forceinline__ void zeros(int k) { for (int i = 0; i < k; ++i) ch('0'); }
};

template <bool WRITE> __device__ void itoa(Sink<WRITE> &o, uint64_t v)
{
    int nd = 1;
    for (uint64_t t = v; t >= 10; t /= 10) ++nd;
    if (WRITE) { uint64_t t = v; for (int i = nd - 1; i >= 0; --i) { o.p[o.n + i] = (char)('0' + (int)(t % 10)); t /= 10; } }
    o.n += nd;
}

// up to 20 decimal digits in nibbles (no private array: nothing of this goes to scratch memory)
struct Digits {
    uint64_t lo = 0; uint32_t hi = 0; int len = 0;
    __device__ __forceinline__ void push(int d) { if (len < 16) lo |= (uint64_t)d << (4 * len); else hi |= (uint32_t)d << (4 * (len - 16)); ++len; }
    __device__ __forceinline__ int at(int i) const { return i < 16 ? (int)((lo >> (4 * i)) & 15u) : (int)((hi >> (4 * (i - 16))) & 15u); }
    __device__ __forceinline__ void decLast() { const int i = len - 1; if (i < 16) lo -= 1ull << (4 * i); else hi -= 1u << (4 * (i - 16)); }   // (the digit is never 0 here)
};

__device__ __forceinline__ DiyFp mul(DiyFp a, DiyFp b)
{
    uint64_t h = __umul64hi(a.f, b.f);
    if ((a.f * b.f) & (1ull << 63)) ++h;
    return {h, a.e + b.e + 64};
}
__device__ __forceinline__ DiyFp normalize(uint64_t f, int e) { const int s = __clzll((long long)f); return {f << s, e - s}; }
__device__ __forceinline__ void grisu_round(Digits &d, uint64_t delta, uint64_t rest, uint64_t tenKappa, uint64_t wpw)
{
    while (rest < wpw && delta - rest >= tenKappa && (rest + tenKappa < wpw || wpw - rest > rest + tenKappa - wpw)) { d.decLast(); rest += tenKappa; }
}
__device__ __forceinline__ int count_digits32(uint32_t n)
{
    int d = 1;
    for (uint32_t lim = 10; d < 10 && n >= lim; lim *= 10) ++d;
    return d;
}

// value > 0, finite: shortest digits and the decimal exponent K (value = digits x 10^K)
__device__ inline void grisu2(double value, Digits &buf, int &K)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(value);
    const int biased = (int)((bits >> 52) & 0x7FF);
    const uint64_t frac = bits & ((1ull << 52) - 1);
    uint64_t f; int e;
    if (biased) { f = frac | (1ull << 52); e = biased - 1075; } else { f = frac; e = -1074; }
    const DiyFp pl = normalize((f << 1) + 1, e - 1);
    uint64_t mf; int me;
    if (f == (1ull << 52)) { mf = (f << 2) - 1; me = e - 2; } else { mf = (f << 1) - 1; me = e - 1; }
    const DiyFp mi = {mf << (me - pl.e), pl.e};
    const double dk = (double)(-61 - pl.e) * 0.30102999566398114 + 347.0;
    int k = (int)dk;
    if ((double)k != dk) ++k;
    const unsigned index = (unsigned)((k >> 3) + 1);
    K = -(-348 + (int)(index << 3));
    const DiyFp c = kPowers[index];
    const DiyFp W = mul(normalize(f, e), c);
    DiyFp Wp = mul(pl, c), Wm = mul(mi, c);
    Wm.f++; Wp.f--;
    uint64_t delta = Wp.f - Wm.f;
    const int sh = -Wp.e;
    const uint64_t oneF = 1ull << sh;
    const uint64_t wpw = Wp.f - W.f;
    uint32_t p1 = (uint32_t)(Wp.f >> sh);
    uint64_t p2 = Wp.f & (oneF - 1);
    int kappa = count_digits32(p1);
    while (kappa > 0) {
        const uint32_t div = kPow10[kappa - 1];
        const uint32_t d = p1 / div;
        p1 %= div;
        if (d || buf.len) buf.push((int)d);
        --kappa;
        const uint64_t tmp = ((uint64_t)p1 << sh) + p2;
        if (tmp <= delta) { K += kappa; grisu_round(buf, delta, tmp, (uint64_t)kPow10[kappa] << sh, wpw); return; }
    }
    for (;;) {
        p2 *= 10; delta *= 10;
        const int d = (int)(p2 >> sh);
        if (d || buf.len) buf.push(d);
        p2 &= oneF - 1;
        --kappa;
        if (p2 < delta) { K += kappa; grisu_round(buf, delta, p2, oneF, wpw * (-kappa < 10 ? kPow10[-kappa] : 0u)); return; }
    }
}

template <bool WRITE> __device__ void dtoa(Sink<WRITE> &o, double value)
{
    if (value != value) { o.lit("NaN"); return; }
    if (value == __longlong_as_double(0x7FF0000000000000ll) || value == __longlong_as_double((long long)0xFFF0000000000000ull)) { o.lit("inf"); return; }
    if (value == 0) { o.lit("0.0"); return; }
    if (value < 0) { o.ch('-'); value = -value; }
    Digits d; int k;
    grisu2(value, d, k);
    const int n = d.len, kk = n + k;
    auto digits = [&](int a, int b) { for (int i = a; i < b; ++i) o.ch((char)('0' + d.at(i))); };
    auto expo = [&](int x) { if (x < 0) { o.ch('-'); x = -x; } itoa(o, (uint64_t)x); };
    if (n <= kk && kk <= 21) { digits(0, n); o.zeros(kk - n); o.lit(".0"); }
    else if (0 < kk && kk <= 21) { digits(0, kk); o.ch('.'); digits(kk, n); }
    else if (-6 < kk && kk <= 0) { o.lit("0."); o.zeros(-kk); digits(0, n); }
    else if (n == 1) { digits(0, 1); o.ch('e'); expo(kk - 1); }
    else { digits(0, 1); o.ch('.'); digits(1, n); o.ch('e'); expo(kk - 1); }
}

// One hit of the JSON / JSON-lines files (Compare.hpp:1640-1668,1757-1800)
template <bool WRITE> __device__ void hit_object(Sink<WRITE> &o, const Args &A, const Entry &h, float best, bool pretty, float coherence)
{
    const char *nm = A.taxNames + A.taxNameOff[h.tax];
    const uint64_t nl = A.taxNameOff[h.tax + 1] - A.taxNameOff[h.tax];
    const float err = (best - h.score) / best;
    if (pretty) {
        o.lit("\t\t\"tax ID\": \""); itoa(o, A.taxIds[h.tax]); o.lit("\",\n\t\t\"Name\": \""); o.bytes(nm, nl);
        o.lit("\",\n\t\t\"k-mer Score\": "); dtoa(o, (double)h.score); o.lit(",\n\t\t\"Relative Score\": "); dtoa(o, h.rel);
        o.lit(",\n\t\t\"Error\": "); dtoa(o, (double)err);
        if (A.coherence) { o.lit(",\n\t\t\"Coherence\": "); dtoa(o, (double)coherence); }
        o.lit("\n\t}");
    } else {
        o.lit(" \"tax ID\": \""); itoa(o, A.taxIds[h.tax]); o.lit("\", \"Name\": \""); o.bytes(nm, nl);
        o.lit("\", \"k-mer Score\": "); dtoa(o, (double)h.score); o.lit(", \"Relative Score\": "); dtoa(o, h.rel);
        o.lit(", \"Error\": "); dtoa(o, (double)err);
        if (A.coherence) { o.lit(",\"Coherence\": "); dtoa(o, (double)coherence); }
        o.ch('}');
    }
}

// The text of read r; returns "contaminated" (--filter, Compare.hpp:1597-1606)
template <bool WRITE> __device__ bool format_read(Sink<WRITE> &o, const Args &A, uint32_t r)
{
    const uint4 m = A.meta[r];
    const Entry *res = A.entries + m.x;
    const int64_t cnt = (int64_t)(m.y & 0x7FFFFFFFu);
    const float maxV = __uint_as_float(m.z);
    const uint64_t number = A.firstRead + r;
    const char *name = A.readNames + A.readNameOff[r];
    const uint64_t nameLen = A.readNameOff[r + 1] - A.readNameOff[r];
    const uint32_t len = A.readLen[r];
    const float coherence = A.coherence ? A.cohScores[r] : 0.f;
    const int64_t beasts = (int64_t)A.beasts;
    if (cnt == 0) {
        switch (A.fmt) {
        case FMT_TSV: itoa(o, number); o.ch('\t'); o.bytes(name, nameLen); if (A.coherence) o.lit("\t-\t-\t-\t-\t-\n"); else o.lit("\t-\t-\t-\t-\n"); break;
        case FMT_JSON:
            if (number == 0) o.lit("{\n"); else o.lit(",\n{\n");
            o.lit("\t\"Read number\": "); itoa(o, number); o.lit(",\n\t\"Specifier from input file\": \""); o.bytes(name, nameLen);
            o.lit("\",\n\t\"Length\": "); itoa(o, len); o.lit(",\n\t\"Top hits\": [\n\t],\n\t\"Further hits\": [\n\t]\n}"); break;
        case FMT_JSONL:
            o.lit("{ \"Read number\": "); itoa(o, number); o.lit(", \"Specifier from input file\": \""); o.bytes(name, nameLen);
            o.lit("\", \"Length\": "); itoa(o, len); o.lit(", \"Top hits\": [], \"Further hits\": [] }\n"); break;
        default: o.lit("U\t"); o.bytes(name, nameLen); o.lit("\t0\t"); o.ch((char)len); o.lit("\tA:00\n"); break;      // one raw byte (Compare.hpp:1568)
        }
        return false;
    }
    const float best = A.best[A.readClass[r]];
    int64_t top = 1;
    for (int64_t i = 1; i < cnt && i < beasts; ++i) { if (res[i].score / maxV > 0.8f) ++top; else break; }
    const bool contaminated = ((double)best - (double)maxV) / (double)best < A.errorThreshold || (A.coherence && coherence >= A.coherenceThreshold);
    float before = 0.f;
    switch (A.fmt) {
    case FMT_TSV: {
        // the four columns are built side by side in the reference; here: how many hits the loop prints, then column by column
        int64_t printed = 0;
        for (int64_t j = 0, i = 0; i < cnt && j < beasts; ++i) { ++printed; if (before != res[i].score) { before = res[i].score; ++j; } }
        uint64_t namesLen = 0;
        for (int64_t i = 0; i < printed; ++i) namesLen += A.taxNameOff[res[i].tax + 1] - A.taxNameOff[res[i].tax];
        if (printed == 0 || (printed == 1 && namesLen == 0)) break;            // (an empty names column: the reference prints nothing)
        itoa(o, number); o.ch('\t'); o.bytes(name, nameLen); o.ch('\t');
        for (int64_t i = 0; i < printed; ++i) { if (i) o.ch(';'); itoa(o, A.taxIds[res[i].tax]); }
        o.ch('\t');
        for (int64_t i = 0; i < printed; ++i) { if (i) o.ch(';'); o.bytes(A.taxNames + A.taxNameOff[res[i].tax], A.taxNameOff[res[i].tax + 1] - A.taxNameOff[res[i].tax]); }
        o.ch('\t');
        for (int64_t i = 0; i < printed; ++i) { if (i) o.ch(';'); dtoa(o, res[i].rel); o.ch(','); dtoa(o, (double)res[i].score); }
        o.ch('\t');
        for (int64_t i = 0; i < printed; ++i) { if (i) o.ch(';'); dtoa(o, (double)((best - res[i].score) / best)); }
        if (A.coherence) { o.ch('\t'); dtoa(o, (double)coherence); }
        o.ch('\n');
    } break;
    case FMT_JSON: {
        if (number == 0) o.lit("{\n"); else o.lit(",\n{\n");
        o.lit("\t\"Read number\": "); itoa(o, number); o.lit(",\n\t\"Specifier from input file\": \""); o.bytes(name, nameLen);
        o.lit("\",\n\t\"Length\": "); itoa(o, len); o.lit(",\n\t\"Top hits\": [\n");
        for (int64_t i = 0; i < top; ++i) { if (i == 0) o.lit("\t{\n"); else o.lit(",\n\t{\n"); hit_object(o, A, res[i], best, true, coherence); }
        o.lit("\n\t],\n\t\"Further hits\": [\n");
        for (int64_t j = top, i = top; i < cnt && j < beasts; ++i) {
            if (j == top) o.lit("\t{\n"); else o.lit(",\n\t{\n");
            hit_object(o, A, res[i], best, true, coherence);
            if (before != res[i].score) { before = res[i].score; ++j; }
        }
        o.lit("\n\t]\n}");
    } break;
    case FMT_JSONL: {
        o.lit("{ \"Read number\": "); itoa(o, number); o.lit(", \"Specifier from input file\": \""); o.bytes(name, nameLen);
        o.lit("\", \"Length\": "); itoa(o, len); o.lit(", \"Top hits\": [");
        for (int64_t i = 0; i < top; ++i) { if (i == 0) o.ch('{'); else o.lit(",{"); hit_object(o, A, res[i], best, false, coherence); }
        o.lit("], \"Further hits\": [");
        for (int64_t j = top, i = top; i < cnt && j < beasts; ++i) {
            if (j == top) o.ch('{'); else o.lit(", {");
            hit_object(o, A, res[i], best, false, coherence);
            if (before != res[i].score) { before = res[i].score; ++j; }
        }
        o.lit("] }\n");
    } break;
    default: {
        o.lit("C\t"); o.bytes(name, nameLen); o.ch('\t'); itoa(o, A.taxIds[res[0].tax]); o.ch('\t'); itoa(o, len); o.ch('\t');
        for (int64_t i = 0; i < top; ++i) { itoa(o, A.taxIds[res[i].tax]); o.ch(':'); dtoa(o, (double)res[i].score); o.ch(' '); }
        for (int64_t j = top, i = top; i < cnt && j < beasts; ++i) {
            itoa(o, A.taxIds[res[i].tax]); o.ch(':'); dtoa(o, (double)res[i].score); o.ch(' ');
            if (before != res[i].score) { before = res[i].score; ++j; }
        }
        o.ch('\n');
    } break;
    }
    return contaminated;
}

// pass 1: bytes of every read's text (+ the contamination flags); pass 2 (below): the text at its offset
__global__ __launch_bounds__(256) void text_size_kernel(Args A, uint64_t *__restrict__ bytes, uint8_t *__restrict__ contaminated)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nReads) return;
    Sink<false> o{nullptr, 0};
    const bool c = format_read<false>(o, A, r);
    bytes[r] = o.n;
    contaminated[r] = c ? 1 : 0;
}
// One wavefront per workgroup writes the text of 64 consecutive reads -- which is ONE contiguous piece of the file -- into LDS
// first (at the same alignment as its place in the buffer) and then copies it out 16 bytes per lane: byte-wise stores
// straight to global memory cost one memory transaction each, and the chip completes about 25 G scattered transactions per
// second whatever their size (tools/scatter_probe.hip) -- 5.5 GB of text took 0.2 s that way.  A piece beyond the LDS
// buffer (reads with very many printed hits) is written directly.
constexpr uint32_t TEXT_LDS = 48u * 1024u;
__global__ __launch_bounds__(64) void text_write_kernel(Args A, const uint64_t *__restrict__ offset, char *__restrict__ text)
{
    __shared__ __attribute__((aligned(16))) char sText[TEXT_LDS];
    const uint32_t r0 = blockIdx.x * 64u, lane = threadIdx.x, r = r0 + lane;
    const uint32_t rEnd = r0 + 64u < A.nReads ? r0 + 64u : A.nReads;
    const uint64_t base = offset[r0], total = offset[rEnd] - base;
    const uint32_t m = (uint32_t)(base & 15u);
    if (total + m > TEXT_LDS) {
        if (r < A.nReads) { Sink<true> o{text + offset[r], 0}; (void)format_read<true>(o, A, r); }
        return;
    }
    if (r < A.nReads) { Sink<true> o{sText + m + (uint32_t)(offset[r] - base), 0}; (void)format_read<true>(o, A, r); }
    __syncthreads();
    char *g0 = text + (base - m);                                       // 16-byte aligned (the buffer is)
    const uint32_t span = m + (uint32_t)total;
    for (uint32_t c = lane * 16u; c < span; c += 64u * 16u) {
        if (c >= m && c + 16u <= span) *reinterpret_cast<uint4 *>(g0 + c) = *reinterpret_cast<const uint4 *>(sText + c);
        else { const uint32_t e = c + 16u < span ? c + 16u : span; for (uint32_t i = c > m ? c : m; i < e; ++i) g0[i] = sText[i]; }
    }
}

// a debugging / test entry: the reference's number formats for arrays of doubles (one value per line)
__global__ void dtoa_probe_kernel(const double *__restrict__ v, uint32_t n, char *__restrict__ out /* 32 bytes per value, zero-terminated */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sink<true> o{out + (size_t)i * 32, 0};
    dtoa<true>(o, v[i]);
    o.ch('\0');
}

} // namespace kasa_text

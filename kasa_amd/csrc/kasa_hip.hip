// kasa_hip.hip -- MI355X (gfx950 / CDNA4) implementation of kASA's `identify` hot path behind the C ABI
// of include/kasa_hip.h.  Written for 64-wide wavefronts and HBM-bound integer work: no MFMA (there is
// no contraction anywhere on this path), coalesced streaming of the sorted query and index arrays,
// LDS for the per-tile scans, everything resident in HBM between stages.
//
// Pipeline of one batch (reference lines in include/kasa_hip.h and DESIGN.md):
//   encode  : reads -> packed 5-bit-letter k-mers; payload = the query's slot in the read-major record array  [encode_kernel]
//   sort    : stable LSD radix sort of (k-mer, slot): five hand-written 8-bit passes over the top 40 key bits
//             (kasa_radix.h) + a rank inside the remaining buckets                                           [pass_kernel, bucket_rank32_kernel]
//   lookup  : two-level prefix table + the tile's index span in LDS: deepest matched level, an index position   [lookup_tile_kernel]
//   group   : per query: flush positions of its levels, their order, taxon segments -> one record, to its slot  [group_kernel]
//   score   : per read, records replayed in the reference's flush order (float32); profile tables in integers   [score_main_kernel,
//             score_other_flat_kernel, row_merge_*, profile_table_kernel; score_kernel for the general case]
//   then, outside the scored path: ranking, per-read text, --coherence                                         [rank_*, text_*, coh_*]
//
// The semantics implemented are the closed form of SURVEY.md section 0.1; tests compare every stage with
// the CPU oracle (oracle/), which is itself pinned to the reference binary's outputs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>                                  // types only: the library is bound at run time (rccl_api below)

#include "../../include/kasa_hip.h"
#include "stdsort_order.h"
#include "kasa_radix.h"
#include "kasa_text.h"

// ------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------
// letters per packed k-mer, key bits and the index meta entry depend on the key width: KeyTraits<Key> below
static constexpr int RANGE_LETTERS = 6;           // depth of the reference's prefix trie (Trie.hpp)
static constexpr int MAX_LEVELS = 25;
static constexpr int TILE = 1024;                 // sorted queries per workgroup in lookup/group
static constexpr int TILE_THREADS = 256;
static constexpr int ITEMS = TILE / TILE_THREADS; // 4
static constexpr uint32_t NOPOS = 0xFFFFFFFFu;
static constexpr int PCAP = 4096;                 // pending (not yet flushed) groups per read (the second pass of the general score kernel: 53 KB of LDS, one wavefront per CU -- rare reads)
static constexpr int TLIST = 256;                // touched taxa of a read kept as a list (else dense scan)

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? KASA_E_NOMEM : KASA_E_HIP, "%s failed: %s",    \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)

extern "C" const char *kasa_last_error(void) { return g_err.c_str(); }

// No C++ exception crosses the C boundary (include/kasa_hip.h): entry points that allocate host memory run under this guard.
#define KASA_GUARDED(call)                                                                          \
    try { return (call); }                                                                          \
    catch (const std::bad_alloc &) { return fail(KASA_E_NOMEM, "host allocation failed"); }         \
    catch (const std::exception &e_) { return fail(KASA_E_ARG, "%s", e_.what()); }                  \
    catch (...) { return fail(KASA_E_HIP, "unexpected exception"); }

extern "C" int kasa_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    if (count) *count = n;
    return KASA_OK;
}

// grow-only device buffer; owns its memory (released with the object that holds it: a context buffer nobody listed cannot leak)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~DevBuf() { release(); }
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return KASA_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 16 + 256;
        static const char *timing = getenv("KASA_ALLOC_TIMING");               // diagnostics: allocations that take long (value: ms, default 20)
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, want);
        if (timing) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double limit = atof(timing) > 0 ? atof(timing) * 1e-3 : 0.02;
            if (dt > limit) fprintf(stderr, "kasa: hipMalloc(%.2f GB) took %.3f s\n", want / 1e9, dt);
        }
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            return fail(KASA_E_NOMEM, "device allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return KASA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
typedef DevBuf ScopedBuf;                                     // a function's own scratch

// ---- a buffer that takes random 32-byte stores is CHOSEN ---------------------------------------------------------------
// The rate at which this chip takes random 32-byte stores is a property of the PHYSICAL memory a buffer got: 21.2, 25.8, 27.2
// or 28.4 G records/s for 34 GB pieces of one device, the same piece always the same and whatever virtual address it is
// mapped at (tools/place_probe.hip, profiles/r06_place_probe.json) -- and the group stage ends in one such store per query:
// its kernels took 56 or 61 ms for the same batch depending on what hipMalloc had handed out (in ONE process before and
// after the record buffer was allocated again; between boxes).  So up to three candidates are allocated (held together: they
// are different memory), each is timed with 2^27 scattered 32-byte stores (5 ms), the first that is fast enough or else
// the best is kept.  Buffers below 8 GB are not worth it (KASA_PLACE_MIN_MB; KASA_PLACE_TRIES=1: plain allocation).
__global__ void place_probe_kernel(uint4 *__restrict__ dst, uint32_t n, unsigned long long slots)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long h = (i + 1ull) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const size_t s = (size_t)__umul64hi(h, slots);                  // uniform over the buffer's 32-byte slots
    dst[s * 2] = make_uint4(i, 0u, 0u, 0u);
    dst[s * 2 + 1] = make_uint4(0u, i, 0u, 0u);
}

struct Placement { uint32_t candidates = 0; float kept = 0.0f, rates[4] = {0.0f, 0.0f, 0.0f, 0.0f}; };

static float place_probe_rate(void *p, size_t bytes, hipStream_t stream)   // G records/s; 0 when it could not be measured
{
    const uint32_t n = 1u << 27;
    const unsigned long long slots = bytes / 32;
    hipEvent_t a = nullptr, b = nullptr;
    if (slots < 2 || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { if (a) (void)hipEventDestroy(a); (void)hipGetLastError(); return 0.0f; }
    float best = 0.0f;
    for (int rep = 0; rep < 3; ++rep) {                              // (the first launch warms up)
        float ms = 0.0f;
        (void)hipEventRecord(a, stream);
        place_probe_kernel<<<n / 256, 256, 0, stream>>>(reinterpret_cast<uint4 *>(p), n, slots);
        (void)hipEventRecord(b, stream);
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess || ms <= 0.0f) { best = 0.0f; (void)hipGetLastError(); break; }
        if (rep) best = std::max(best, (float)(n / (ms * 1e6)));
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best;
}

static int reserve_placed(DevBuf &buf, size_t bytes, hipStream_t stream, Placement *out)
{
    if (bytes <= buf.cap) return KASA_OK;
    const char *minEnv = getenv("KASA_PLACE_MIN_MB"), *triesEnv = getenv("KASA_PLACE_TRIES");
    const size_t minBytes = (size_t)(minEnv ? atol(minEnv) : 8192) << 20;
    const int tries = std::min(4, triesEnv ? atoi(triesEnv) : 3);
    if (tries <= 1 || bytes < minBytes) return buf.reserve(bytes);
    const float goodEnough = 25.0f;                                  // G records/s by THIS probe (2^27 stores: 20.5 on the slow kind of memory, 23.7, 25.7 on the fast ones)
    buf.release();
    DevBuf cand[4];
    Placement pl;
    int best = -1;
    for (int t = 0; t < tries; ++t) {
        if (t > 0) {                                                 // a further candidate only where a quarter of the device stays free beside it
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); break; }
            if (freeB < bytes + bytes / 16 + totalB / 4) break;
        }
        const int rc = cand[t].reserve(bytes);
        if (rc) { if (t == 0) return rc; break; }
        pl.rates[t] = place_probe_rate(cand[t].p, bytes, stream);
        pl.candidates = (uint32_t)t + 1;
        if (best < 0 || pl.rates[t] > pl.rates[best]) best = t;
        if (pl.rates[t] >= goodEnough || pl.rates[t] == 0.0f) break; // (not measurable: no reason to go on)
    }
    pl.kept = pl.rates[best];
    buf = std::move(cand[best]);                                     // (the others go back as `cand` leaves)
    if (getenv("KASA_PLACE_VERBOSE"))
        fprintf(stderr, "kasa: buffer of %.1f GB for random 32-byte stores: %u candidate(s) %.1f %.1f %.1f %.1f -> kept %.1f G records/s\n", bytes / 1e9, pl.candidates,
                pl.rates[0], pl.rates[1], pl.rates[2], pl.rates[3], pl.kept);
    if (out) *out = pl;
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// Keys: uint64_t for the 64-bit index (12 letters, 60 bits) or key128 for the 128-bit index (25 letters, 125 bits;
// source/utils/uint128_t.hpp).  Every key-typed kernel is a template instantiated for both.
typedef unsigned __int128 key128;
template <class Key> struct KeyTraits;
template <> struct KeyTraits<uint64_t> {
    static constexpr int LETTERS = 12, BITS = 60, SHIFT = 4;
    typedef uint8_t Meta;                              // low nibble: letters shared with the previous entry, high nibble: dupLvl
    static constexpr int META_SHIFT = 4, META_MASK = 15;
};
template <> struct KeyTraits<key128> {
    static constexpr int LETTERS = 25, BITS = 125, SHIFT = 3;
    typedef uint16_t Meta;                             // the same two counts, a byte each
    static constexpr int META_SHIFT = 8, META_MASK = 255;
};

__host__ __device__ __forceinline__ int clz_key(uint64_t x) { return __builtin_clzll(x); }             // x != 0
__host__ __device__ __forceinline__ int clz_key(key128 x)
{
    const uint64_t hi = (uint64_t)(x >> 64);
    return hi ? __builtin_clzll(hi) : 64 + __builtin_clzll((uint64_t)x);
}

template <class Key> __device__ __forceinline__ int lcp_of_xor(Key x)   // letters two keys share, from their XOR
{
    x <<= KeyTraits<Key>::SHIFT;
    return x ? (clz_key(x) / 5) : KeyTraits<Key>::LETTERS;
}
template <class Key> __device__ __forceinline__ int lcp_letters(Key a, Key b) { return lcp_of_xor<Key>(a ^ b); }

// 5-bit-field constants over the letters of a key: '^' (30) in every field, the low four bits, bit 4
template <class Key> __host__ __device__ constexpr Key field_repeat(unsigned v)
{
    Key r = 0;
    for (int i = 0; i < KeyTraits<Key>::LETTERS; ++i) r |= (Key)v << (5 * i);
    return r;
}

__device__ __forceinline__ int group_letters(int k) { return k < RANGE_LETTERS ? RANGE_LETTERS : k; }

// Running sums over the 64 lanes of a wavefront by DPP (data-parallel primitives: a lane reads its neighbour's register
// inside the VALU instruction): four shifts inside the rows of 16 lanes, then the last lane of a row added to the rows
// behind it.  Twelve instructions and no trip through the LDS crossbar, where the __shfl_up form (ds_bpermute) takes six
// dependent LDS round trips.  ALL 64 lanes must be active.
#define KASA_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    v += (uint32_t)KASA_DPP(v, 0x111, 0xf);                           // row_shr:1
    v += (uint32_t)KASA_DPP(v, 0x112, 0xf);                           // row_shr:2
    v += (uint32_t)KASA_DPP(v, 0x114, 0xf);                           // row_shr:4
    v += (uint32_t)KASA_DPP(v, 0x118, 0xf);                           // row_shr:8
    v += (uint32_t)KASA_DPP(v, 0x142, 0xa);                           // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)KASA_DPP(v, 0x143, 0xc);                           // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_total(uint32_t inclusive) { return (uint32_t)__builtin_amdgcn_readlane((int)inclusive, 63); }
// one lane's value for all (the lane number is a constant: v_readlane, no LDS round trip as with __shfl)
template <int LANE> __device__ __forceinline__ uint32_t lane_value(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, LANE); }
template <int LANE> __device__ __forceinline__ unsigned long long lane_value(unsigned long long v)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), LANE) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, LANE);
}
// the value of the lane before (lane 0: `first`)
__device__ __forceinline__ int lane_before(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1
// the same, restarting at every lane whose flag is set: on return f = "a flag was set at or before this lane"
__device__ __forceinline__ void wave_seg_incl_sum(uint32_t &v, bool &f)
{
    uint32_t fl = f ? 1u : 0u;
#define KASA_SEG_STEP(ctrl, rows) do { const uint32_t ov = (uint32_t)KASA_DPP(v, ctrl, rows), of = (uint32_t)KASA_DPP(fl, ctrl, rows); v += fl ? 0u : ov; fl |= of; } while (0)
    KASA_SEG_STEP(0x111, 0xf); KASA_SEG_STEP(0x112, 0xf); KASA_SEG_STEP(0x114, 0xf); KASA_SEG_STEP(0x118, 0xf);
    KASA_SEG_STEP(0x142, 0xa); KASA_SEG_STEP(0x143, 0xc);
#undef KASA_SEG_STEP
    f = fl != 0u;
}

// ------------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------------
struct kasa_index {
    int device = 0;
    uint64_t n = 0;
    uint32_t nTaxa = 0;
    bool wide = false; // 128-bit keys (K = 25 letters)
    int letters() const { return wide ? 25 : 12; }
    DevBuf kmer;   // u64[n] or key128[n]
    DevBuf tax;    // u32[n] dense taxon index
    DevBuf meta;   // KeyTraits::Meta[n]: letters shared with the previous entry | letters shared with the nearest
                   //        earlier entry of the same taxon
    DevBuf table;  // u32[2^tb]: number of entries whose top `tb` key bits are <= b
    int tb = 0;
    uint64_t bytes() const { return kmer.cap + tax.cap + meta.cap + table.cap; }
};

template <class Key>
__global__ void unpack_records_kernel(const uint8_t *__restrict__ rec, uint64_t n, Key *__restrict__ kmer,
                                      uint32_t *__restrict__ taxid)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (sizeof(Key) == 8) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(rec + i * 12); // 12-byte records are 4-aligned
        kmer[i] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        taxid[i] = w[2];
    } else {   // packedLargePair (packedPairs.hpp:132-155): low word, high word, tax id = 20 bytes, 4-aligned
        const uint32_t *w = reinterpret_cast<const uint32_t *>(rec + i * 20);
        const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        kmer[i] = ((key128)hi << 64) | lo;
        taxid[i] = w[4];
    }
}

__global__ void dense_tax_kernel(uint32_t *__restrict__ tax, uint64_t n, const uint32_t *__restrict__ sortedIds,
                                 const uint32_t *__restrict__ denseOf, uint32_t nTaxa, uint32_t *__restrict__ bad)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = tax[i];
    uint32_t lo = 0, hi = nTaxa;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sortedIds[mid] < id) lo = mid + 1; else hi = mid;
    }
    if (lo < nTaxa && sortedIds[lo] == id) tax[i] = denseOf[lo];
    else { tax[i] = 0; atomicAdd(bad, 1u); }
}

template <class Key>
__global__ void index_check_kernel(const Key *__restrict__ kmer, const uint32_t *__restrict__ taxid, uint64_t n,
                                   uint32_t *__restrict__ bad)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i == 0 || i >= n) return;
    const Key a = kmer[i - 1], b = kmer[i];
    if (a > b || (a == b && taxid[i - 1] >= taxid[i]) || (b >> KeyTraits<Key>::BITS)) atomicAdd(bad, 1u);
}

template <class Key>
__global__ void lcp_prev_kernel(const Key *__restrict__ kmer, uint64_t n, typename KeyTraits<Key>::Meta *__restrict__ meta)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    meta[i] = (i == 0) ? 0 : (typename KeyTraits<Key>::Meta)lcp_letters<Key>(kmer[i - 1], kmer[i]);
}

// after a stable sort of positions by taxon: consecutive positions of one taxon are index neighbours
// of that taxon, so the letters they share is exactly "how deep the earlier one shadows the later one"
template <class Key>
__global__ void dup_level_kernel(const uint32_t *__restrict__ taxSorted, const uint32_t *__restrict__ order,
                                 const Key *__restrict__ kmer, uint64_t n, typename KeyTraits<Key>::Meta *__restrict__ meta)
{
    typedef KeyTraits<Key> T;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0;
    if (i > 0 && taxSorted[i] == taxSorted[i - 1]) d = (uint32_t)lcp_letters<Key>(kmer[order[i - 1]], kmer[order[i]]);
    const uint32_t pos = order[i];
    meta[pos] = (typename T::Meta)((meta[pos] & T::META_MASK) | (d << T::META_SHIFT));
}

template <class Key>
__global__ void table_mark_kernel(const Key *__restrict__ kmer, uint64_t n, int shift, uint32_t *__restrict__ table)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = (uint64_t)(kmer[i] >> shift);
    if (i + 1 == n || (uint64_t)(kmer[i + 1] >> shift) != b) table[b] = (uint32_t)(i + 1);
}

template <class Key>
__global__ void trie_check_kernel(const uint32_t *__restrict__ prefix, const uint64_t *__restrict__ start,
                                  const uint64_t *__restrict__ count, uint64_t nTrie, const Key *__restrict__ kmer,
                                  uint64_t n, uint32_t *__restrict__ bad)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= nTrie) return;
    const uint64_t s = start[i], c = count[i];
    const int sh = 5 * (KeyTraits<Key>::LETTERS - RANGE_LETTERS);
    bool ok = c > 0 && s + c <= n;
    if (ok) ok = (uint64_t)(kmer[s] >> sh) == prefix[i] && (uint64_t)(kmer[s + c - 1] >> sh) == prefix[i];
    if (ok && s > 0) ok = (uint64_t)(kmer[s - 1] >> sh) < prefix[i];
    if (ok && s + c < n) ok = (uint64_t)(kmer[s + c] >> sh) > prefix[i];
    if (!ok) atomicAdd(bad, 1u);
}

static inline unsigned blocks_for(uint64_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

template <class Key>
static int index_create_impl(int device, const void *records, uint64_t nRecords,
                             const uint32_t *triePrefix, const uint64_t *trieCount, uint64_t nTrie,
                             const uint32_t *taxIds, uint32_t nTaxa, kasa_index **out);

extern "C" int kasa_index_create(int device, const void *records, uint64_t nRecords, int recordBytes,
                                 const uint32_t *triePrefix, const uint64_t *trieCount, uint64_t nTrie,
                                 const uint32_t *taxIds, uint32_t nTaxa, kasa_index **out)
{
    if (!out) return fail(KASA_E_ARG, "kasa_index_create: out is NULL");
    *out = nullptr;
    if (recordBytes == 6) {
        // "halved" index of shrink strategy 2 (source/modes/Shrink.hpp:78-143, source/utils/packedPairs.hpp:100-105):
        // {u32 low 30 bits of the k-mer, u16 dense taxon index}; the upper 30 bits are the entry's `_trie` prefix.
        // Rebuilt here into full records, then loaded like any other index.
        if (!records || !triePrefix || !trieCount || !nTrie || !taxIds) return fail(KASA_E_ARG, "kasa_index_create: a halved index needs its trie and content mapping");
        std::vector<uint8_t> full;
        try { full.resize((size_t)nRecords * 12); } catch (...) { return fail(KASA_E_NOMEM, "host allocation failed"); }
        const uint8_t *src = static_cast<const uint8_t *>(records);
        uint64_t i = 0;
        for (uint64_t t = 0; t < nTrie; ++t)
            for (uint64_t c2 = 0; c2 < trieCount[t]; ++c2, ++i) {
                if (i >= nRecords) return fail(KASA_E_ARG, "kasa_index_create: the trie file counts more entries than the index has");
                uint32_t low; uint16_t tix;
                memcpy(&low, src + i * 6, 4); memcpy(&tix, src + i * 6 + 4, 2);
                if (tix >= nTaxa) return fail(KASA_E_ARG, "kasa_index_create: halved index entry %llu names taxon index %u of %u", (unsigned long long)i, tix, nTaxa);
                const uint64_t km = ((uint64_t)triePrefix[t] << 30) | (low & 0x3FFFFFFFu);
                const uint32_t tid = taxIds[tix];
                memcpy(&full[i * 12], &km, 8); memcpy(&full[i * 12 + 8], &tid, 4);
            }
        if (i != nRecords) return fail(KASA_E_ARG, "kasa_index_create: the trie file counts %llu entries, the index has %llu", (unsigned long long)i, (unsigned long long)nRecords);
        return kasa_index_create(device, full.data(), nRecords, 12, triePrefix, trieCount, nTrie, taxIds, nTaxa, out);
    }
    if (recordBytes == 20) { KASA_GUARDED(index_create_impl<key128>(device, records, nRecords, triePrefix, trieCount, nTrie, taxIds, nTaxa, out)) }
    if (recordBytes == 12) { KASA_GUARDED(index_create_impl<uint64_t>(device, records, nRecords, triePrefix, trieCount, nTrie, taxIds, nTaxa, out)) }
    return fail(KASA_E_ARG, "kasa_index_create: records are 12 bytes {u64 kmer, u32 taxid}, 20 bytes {u64 low, u64 high, u32 taxid} (k <= 25) or 6 bytes (halved), got %d", recordBytes);
}

template <class Key>
static int index_create_impl(int device, const void *records, uint64_t nRecords,
                             const uint32_t *triePrefix, const uint64_t *trieCount, uint64_t nTrie,
                             const uint32_t *taxIds, uint32_t nTaxa, kasa_index **out)
{
    typedef KeyTraits<Key> T;
    typedef typename T::Meta Meta;
    constexpr size_t REC = sizeof(Key) + 4;
    if (!records && nRecords) return fail(KASA_E_ARG, "kasa_index_create: records is NULL");
    if (nRecords == 0) return fail(KASA_E_ARG, "The index file cannot be found or is empty!");
    if (nRecords >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_index_create: %llu records exceed the 32-bit position range of this build", (unsigned long long)nRecords);
    if (!taxIds || nTaxa < 2) return fail(KASA_E_ARG, "kasa_index_create: content mapping missing");
    if (nTaxa >= (1u << 22)) return fail(KASA_E_LIMIT, "kasa_index_create: %u taxa exceed the 2^22 taxon indices an event record can name", nTaxa);
    int ndev = 0;
    kasa_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(KASA_E_HIP, "kasa_index_create: no HIP device %d (found %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    (void)hipGetLastError(); // do not inherit a stale sticky error from unrelated earlier calls

    kasa_index *ix = new (std::nothrow) kasa_index();
    if (!ix) return fail(KASA_E_NOMEM, "host allocation failed");
    ix->device = device; ix->n = nRecords; ix->nTaxa = nTaxa; ix->wide = sizeof(Key) > 8;
    int rc = KASA_OK;
    DevBuf raw, ids, dense, bad, order, taxSorted, iota, tmp, tpre, tstart, tcount;
    auto cleanup = [&](int code) {
        raw.release(); ids.release(); dense.release(); bad.release(); order.release(); taxSorted.release();
        iota.release(); tmp.release(); tpre.release(); tstart.release(); tcount.release();
        if (code != KASA_OK) { ix->kmer.release(); ix->tax.release(); ix->meta.release(); ix->table.release(); delete ix; }
        return code;
    };
#define TRY(x) do { rc = (x); if (rc != KASA_OK) return cleanup(rc); } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return cleanup(fail(e_ == hipErrorOutOfMemory ? KASA_E_NOMEM : KASA_E_HIP, "%s failed: %s", #x, hipGetErrorString(e_))); } while (0)
    const uint64_t n = nRecords;
    TRY(raw.reserve(n * REC));
    TRY(ix->kmer.reserve(n * sizeof(Key)));
    TRY(ix->tax.reserve(n * 4));
    TRY(ix->meta.reserve(n * sizeof(Meta)));
    TRY(bad.reserve(16));
    TRYHIP(hipMemcpy(raw.p, records, n * REC, hipMemcpyDefault));        // host memory (e.g. the mmap'ed index file) or device memory
    TRYHIP(hipMemset(bad.p, 0, 16));
    unpack_records_kernel<Key><<<blocks_for(n, 256), 256>>>(raw.as<uint8_t>(), n, ix->kmer.as<Key>(), ix->tax.as<uint32_t>());
    index_check_kernel<Key><<<blocks_for(n, 256), 256>>>(ix->kmer.as<Key>(), ix->tax.as<uint32_t>(), n, bad.as<uint32_t>());
    raw.release();
    // taxid -> dense index (Compare.hpp:139-143)
    {
        std::vector<uint32_t> perm(nTaxa), sid(nTaxa), did(nTaxa);
        for (uint32_t i = 0; i < nTaxa; ++i) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return taxIds[a] < taxIds[b]; });
        for (uint32_t i = 0; i < nTaxa; ++i) { sid[i] = taxIds[perm[i]]; did[i] = perm[i]; }
        TRY(ids.reserve(nTaxa * 4));
        TRY(dense.reserve(nTaxa * 4));
        TRYHIP(hipMemcpy(ids.p, sid.data(), nTaxa * 4, hipMemcpyHostToDevice));
        TRYHIP(hipMemcpy(dense.p, did.data(), nTaxa * 4, hipMemcpyHostToDevice));
        dense_tax_kernel<<<blocks_for(n, 256), 256>>>(ix->tax.as<uint32_t>(), n, ids.as<uint32_t>(), dense.as<uint32_t>(), nTaxa, bad.as<uint32_t>() + 1);
    }
    uint32_t hbad[4] = {0, 0, 0, 0};
    TRYHIP(hipMemcpy(hbad, bad.p, 16, hipMemcpyDeviceToHost));
    if (hbad[0]) return cleanup(fail(KASA_E_ARG, "kasa_index_create: the index is not sorted by (kmer, taxid), not unique, or uses more than %d key bits (%u violations)", T::BITS, hbad[0]));
    if (hbad[1]) return cleanup(fail(KASA_E_ARG, "kasa_index_create: %u index entries carry a tax ID the content file does not know", hbad[1]));

    // meta: shared letters with the previous entry and with the previous entry of the same taxon
    lcp_prev_kernel<Key><<<blocks_for(n, 256), 256>>>(ix->kmer.as<Key>(), n, ix->meta.as<Meta>());
    {
        TRY(order.reserve(n * 4));
        TRY(taxSorted.reserve(n * 4));
        size_t tmpBytes = 0;
        unsigned bits = 1;
        while ((1ull << bits) < nTaxa) ++bits;
        rocprim::counting_iterator<uint32_t> cnt(0);
        TRYHIP(rocprim::radix_sort_pairs(nullptr, tmpBytes, ix->tax.as<uint32_t>(), taxSorted.as<uint32_t>(), cnt,
                                         order.as<uint32_t>(), (size_t)n, 0u, bits, (hipStream_t)0));
        TRY(tmp.reserve(tmpBytes));
        TRYHIP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, ix->tax.as<uint32_t>(), taxSorted.as<uint32_t>(), cnt,
                                         order.as<uint32_t>(), (size_t)n, 0u, bits, (hipStream_t)0));
        dup_level_kernel<Key><<<blocks_for(n, 256), 256>>>(taxSorted.as<uint32_t>(), order.as<uint32_t>(), ix->kmer.as<Key>(), n, ix->meta.as<Meta>());
        TRYHIP(hipDeviceSynchronize());
        order.release(); taxSorted.release();
    }
    // two-level prefix table: level 1 = top `tb` key bits -> entry range, level 2 = binary search
    {
        int tb = 8;
        while (tb < 28 && (1ull << (tb + 1)) <= n) ++tb;
        ix->tb = tb;
        const uint64_t nb = 1ull << tb;
        TRY(ix->table.reserve(nb * 4));
        TRYHIP(hipMemset(ix->table.p, 0, nb * 4));
        table_mark_kernel<Key><<<blocks_for(n, 256), 256>>>(ix->kmer.as<Key>(), n, T::BITS - tb, ix->table.as<uint32_t>());
        size_t tmpBytes = 0;
        TRYHIP(rocprim::inclusive_scan(nullptr, tmpBytes, ix->table.as<uint32_t>(), ix->table.as<uint32_t>(), (size_t)nb, rocprim::maximum<uint32_t>(), (hipStream_t)0));
        TRY(tmp.reserve(tmpBytes));
        TRYHIP(rocprim::inclusive_scan(tmp.p, tmpBytes, ix->table.as<uint32_t>(), ix->table.as<uint32_t>(), (size_t)nb, rocprim::maximum<uint32_t>(), (hipStream_t)0));
    }
    // the reference's `_trie` file, when given, must describe the same ranges (Trie.hpp:398-462)
    if (triePrefix && trieCount && nTrie) {
        std::vector<uint64_t> start(nTrie);
        uint64_t run = 0;
        for (uint64_t i = 0; i < nTrie; ++i) { start[i] = run; run += trieCount[i]; }
        if (run != n) return cleanup(fail(KASA_E_ARG, "kasa_index_create: the trie file counts %llu entries, the index has %llu", (unsigned long long)run, (unsigned long long)n));
        TRY(tpre.reserve(nTrie * 4)); TRY(tstart.reserve(nTrie * 8)); TRY(tcount.reserve(nTrie * 8));
        TRYHIP(hipMemcpy(tpre.p, triePrefix, nTrie * 4, hipMemcpyHostToDevice));
        TRYHIP(hipMemcpy(tstart.p, start.data(), nTrie * 8, hipMemcpyHostToDevice));
        TRYHIP(hipMemcpy(tcount.p, trieCount, nTrie * 8, hipMemcpyHostToDevice));
        TRYHIP(hipMemset(bad.p, 0, 16));
        trie_check_kernel<Key><<<blocks_for(nTrie, 256), 256>>>(tpre.as<uint32_t>(), tstart.as<uint64_t>(), tcount.as<uint64_t>(), nTrie, ix->kmer.as<Key>(), n, bad.as<uint32_t>());
        TRYHIP(hipMemcpy(hbad, bad.p, 16, hipMemcpyDeviceToHost));
        if (hbad[0]) return cleanup(fail(KASA_E_ARG, "kasa_index_create: the trie file does not match the index (%u ranges differ)", hbad[0]));
    }
    TRYHIP(hipDeviceSynchronize());
    TRYHIP(hipGetLastError());
#undef TRY
#undef TRYHIP
    cleanup(KASA_OK);
    *out = ix;
    return KASA_OK;
}

extern "C" void kasa_index_destroy(kasa_index *ix)
{
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    ix->kmer.release(); ix->tax.release(); ix->meta.release(); ix->table.release();
    delete ix;
}

extern "C" uint64_t kasa_index_size(const kasa_index *ix) { return ix ? ix->n : 0; }
extern "C" uint64_t kasa_index_device_bytes(const kasa_index *ix) { return ix ? ix->bytes() : 0; }

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct StageTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> open;
    std::vector<hipEvent_t> pool;
    double ms = 0.0;
    uint64_t launches = 0;
};

struct kasa_ctx {
    const kasa_index *ix = nullptr;
    int device = 0;
    int kHigh = 12, kLow = 7, nK = 6, frames = 3;
    bool protein = false;                      // amino-acid input (kasa_ctx_set_protein)
    int enc_mode() const { return protein ? 2 : (frames == 1 ? 1 : 0); }   // ENC_PROTEIN / ENC_ONE / ENC_DNA
    int strands() const { return (frames == 6 && !protein) ? 2 : 1; }      // kASA.hpp:181: protein input switches --six off
    hipStream_t stream = nullptr;
    DevBuf scanTmp;
    // batch state
    int64_t nReads = 0;
    uint64_t nBases = 0;
    uint64_t nQ = 0;
    uint32_t maxCnt = 0;
    int state = 0; // 0 none, 1 uploaded, 2 encoded, 3 sorted+lookup, 4 scored
    bool haveScores = false;
    bool csrPacked = false;                    // the rows of this batch have been packed into outTax / outScore (kasa_batch_scores_fetch)
    DevBuf rankDen, rankClass, rankMeta, rankOut, rankList, rankScratch; uint64_t rankCap = 0, rankEntries = 0;   // kasa_batch_rank
    bool rankValid = false; uint32_t rankFlagged = 0;          // ... of THIS batch; reads it left to the host
    uint32_t rankClasses = 0;                                  // denominator rows kasa_batch_rank was given (kasa_batch_text's bestScore has as many)
    DevBuf taxText, taxTextOff, taxTextIds;                    // kasa_ctx_set_taxa_text: names back to back, u64[nTaxa + 1], u32[nTaxa]
    bool haveTaxText = false;
    DevBuf txtNames, txtNameOff, txtLen, txtBest, txtBytes, txtOff, txtOut, txtFlags;   // kasa_batch_text
    uint64_t txtTotal = 0; bool txtValid = false;
    const float *cohScores = nullptr;                          // device: the scores of the last kasa_batch_coherence of this batch
    bool grouped = false; uint32_t poolUsed = 1; // event records + pool of this batch are in place (group stage or import)
    bool groupCoop = false;                     // group_kernel's cooperative form (long taxon lists were met; sticky)
    bool recSorted = false;                     // ... in sorted order (exported for another rank), not in their slots
    uint32_t *recOut = nullptr;                 // ... written straight into the caller's buffer (kasa_batch_group_to), not into `rec`
    int recWords() const { return nK <= 8 ? 8 : 16; }   // RecTraits: 32-byte records up to 8 levels, 64-byte ones up to 25
    // words between two records in `rec` (the CELL): recWords(), or 16 for narrow records that group2_kernel stores as whole
    // 64-byte cells -- a random 32-byte store is a partial write of a cell and leaves at 27 G records/s, a whole cell written
    // by four lanes at 77 G (tools/scatter_probe.hip, profiles/r06_scatter_probe.json); the readers then stream cells
    uint32_t recCW = 8;
    bool noWideCells = false;                   // the device has no room for 64-byte cells (sticky)
    // buffers
    DevBuf lut, bases, baseOff, kmerOff;       // u8[366], u8[], i64[nSeq+1], u64[nReads+1] (k-mers per READ, running sum)
    const uint8_t *basesPtr = nullptr;         // the batch's bases: bases.p, or the caller's own device memory (kasa_batch_upload_device)
    DevBuf seqOff, seqRead;                    // u64[nSeq+1] k-mer offset of every uploaded sequence, u32[nSeq] its read
    int64_t nSeq = 0; bool haveSeqRead = false;
    DevBuf qKmerA, qKmerB, qReadA, qReadB;     // double buffers of the query arrays
    DevBuf sortBig;                            // heads / begins / ends of the long buckets (sort_and_range)
    DevBuf depth, rep;                         // u8[nQ], u32[nQ]
    DevBuf tileFirst, tileNext, tileBounds;    // u32[nK][nTiles]; index span of every tile
    DevBuf rTab;                               // floor(2^64 / n) for n < 8192 (profile_group_accum_kernel)
    DevBuf tileList;                           // tiles group2_kernel leaves to group_kernel (long lists, walks beyond the staged span)
    Placement recPlacement;                           // how the record buffer was chosen (reserve_placed)
    uint32_t lastSlowTiles = 0, lastSlowTiles2 = 0;   // tiles group2_kernel listed; of those, tiles its second chance (larger park buffer) listed again
    DevBuf tileChunks;                         // tile_suffix: minima of chunks of 1024 tiles
    int lookupMode = 0;                        // 0 = streaming tiles, 1 = per-query search only
    DevBuf rec;                                // event records, recWords() u32 each, by slot
    DevBuf pool, plist, sortTmp, misc;         // taxon segment lists, positions by read (slot fix-up), rocPRIM temp, counters
    DevBuf recIn;                              // imported records in sorted order, before they go to their slots
    DevBuf slotBuf;                            // u32[nQ]: slot of every sorted position when it does not come out of the sort
    const uint32_t *slotOf = nullptr;          // slot of sorted position p (payload of the sort, or slotBuf); NULL: not known yet
    bool payloadIsSlot = false;                // what the encoder gave the sort as payload: slots (ranked reads) or read ids
    DevBuf flushOff, flushPos, flushOff2, flushPos2;   // general score kernel: flush positions of the listed reads' queries
    DevBuf ovList;                             // reads the first general pass hands to the second
    DevBuf fbList2;                            // reads score_dense_kernel hands to the general kernel
    uint32_t lastDenseReads = 0;               // reads of the last batch score_dense_kernel kept
    DevBuf ovList2, gwin;                      // ... the second to the third (narrow records); the third pass's pending windows
    uint32_t lastThirdPassReads = 0;
    // very long reads: events sorted by (read, taxon, flush position, level), float chains per (read, taxon) (kasa_replay.h)
    DevBuf encLong;                            // sequences the encoder spreads over all wavefronts (count, list)
    DevBuf wireOff; uint64_t wireQueries = 0, wireWords = 0; const uint32_t *wireRecords = nullptr;   // kasa_batch_records_pack_size -> _pack: word offsets of the blocks
    DevBuf esrLong, esrShort, esrIota, esrQOff, esrReadEv, esrEvCnt, esrEvOff, esrKeyA, esrKeyB, esrValA, esrValB, esrChain, esrChainScore, esrBig;
    uint32_t lastReplayReads = 0; uint64_t lastReplayEvents = 0;
    uint32_t lastOverflowReads = 0;
    DevBuf scratch, touched, fbList, fastScratch, profKeys, profSorted, profSorted2;   // per-block dense score rows; reads left to the slow kernel
    uint64_t profLeftHint = 0;                  // per-level keys the last batch's table pass left over, when the list was too short for them
    bool forceSlowScore = false; uint32_t lastSlowReads = 0; int debugFlags = 0;
    DevBuf rowPos, rowLen, rowKey, rowOff, st, outTax, outScore;
    DevBuf cntUnique, cntTotal, cntAllHi, cntAllMid, cntAllLo; // u64[nK*nTaxa] each
    uint64_t poolCap = 0, stCap = 0, keyCap = 0, keyCapScore = 0, nnz = 0;   // keyCap: group keys (narrow records); keyCapScore: the per-read side's keys (wide records)
    void *qKmer = nullptr; uint32_t *qRead = nullptr; // current (valid) query arrays; keys are u64 or key128 as the index
    size_t keyBytes() const { return ix->wide ? 16 : 8; }
    int K() const { return ix->letters(); }
    template <class Key> Key *keys() const { return static_cast<Key *>(qKmer); }
    StageTimer timers[KASA_STAGE_COUNT];
    StageTimer kernels[KASA_KERNEL_COUNT];      // single kernels timed alone (kasa_ctx_kernel_ms)
    uint64_t lastStaged = 0, lastKeys = 0, lastContrib = 0;   // of the last batch: staging records, profile keys, (event, taxon) contributions
    DevBuf rawOff;                             // the caller's sequence offsets as uploaded
    DevBuf cohLen, cohState;                   // kasa_batch_coherence: match length of every emitted k-mer; walk state per chunk of reads + scores
    uint64_t nEmitted = 0; bool uniqueDone = false, readsUploaded = false;   // k-mers the encoder emitted (nQ shrinks with -e); -e was applied to this batch
    // every device buffer of the context: what kasa_ctx_destroy releases and kasa_ctx_device_bytes adds up (ONE list)
    std::vector<DevBuf *> buffers()
    {
        return {&lut, &bases, &baseOff, &kmerOff, &seqOff, &seqRead, &qKmerA, &qKmerB, &qReadA, &qReadB, &depth, &rep, &tileFirst, &tileNext, &tileBounds, &tileList, &rTab,
                &tileChunks, &rec, &pool, &plist, &sortTmp, &slotBuf, &recIn, &flushOff, &flushPos, &flushOff2, &flushPos2, &misc, &scratch, &ovList, &ovList2, &fbList2,
                &gwin, &touched, &fbList, &fastScratch, &profKeys, &profSorted, &profSorted2, &rowPos, &rowLen, &rowKey, &rowOff, &st, &cntAllMid, &outTax,
                &outScore, &cntUnique, &cntTotal, &cntAllHi, &cntAllLo, &rawOff, &cohLen, &cohState, &sortBig, &rankDen, &rankClass, &rankMeta, &rankOut,
                &rankList, &rankScratch, &scanTmp, &taxText, &taxTextOff, &taxTextIds, &txtNames, &txtNameOff, &txtLen, &txtBest, &txtBytes, &txtOff, &txtOut,
                &txtFlags, &encLong, &wireOff, &esrLong, &esrShort, &esrIota, &esrQOff, &esrReadEv, &esrEvCnt, &esrEvOff, &esrKeyA, &esrKeyB, &esrValA, &esrValB, &esrChain,
                &esrChainScore, &esrBig};
    }
};

static int timer_begin(kasa_ctx *c, StageTimer &t, hipEvent_t *a, hipEvent_t *b)
{
    auto get = [&](hipEvent_t *e) -> int {
        if (!t.pool.empty()) { *e = t.pool.back(); t.pool.pop_back(); return KASA_OK; }
        HIPCHK(hipEventCreate(e));
        return KASA_OK;
    };
    int rc = get(a); if (rc) return rc;
    rc = get(b); if (rc) return rc;
    HIPCHK(hipEventRecord(*a, c->stream));
    return KASA_OK;
}

static int timer_end(kasa_ctx *c, StageTimer &t, hipEvent_t a, hipEvent_t b)
{
    HIPCHK(hipEventRecord(b, c->stream));
    t.open.emplace_back(a, b);
    t.launches++;
    return KASA_OK;
}

static int timer_resolve(StageTimer &t)
{
    for (auto &pr : t.open) {
        HIPCHK(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        t.ms += ms;
        t.pool.push_back(pr.first);
        t.pool.push_back(pr.second);
    }
    t.open.clear();
    return KASA_OK;
}

static void builtin_codon_table(uint8_t lut[366])
{
    // kASA.hpp:621-667: standard code, stops TAA/TAG -> '[', TGA -> ']', any Z -> '_', else any X -> '^';
    // index = b0*64 + b1*8 + b2 with b = (c & 14) >> 1 (A,C,T,G,X,Z = 0..5), value = letter & 31
    static const char aaTCAG[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const int tcag[4] = {2, 1, 0, 3};
    memset(lut, 0, 366);
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b)
            for (int c = 0; c < 6; ++c) {
                char aa;
                if (a == 5 || b == 5 || c == 5) aa = '_';
                else if (a == 4 || b == 4 || c == 4) aa = '^';
                else {
                    const int i = tcag[a] * 16 + tcag[b] * 4 + tcag[c];
                    aa = aaTCAG[i];
                    if (aa == '*') aa = (i == 14) ? ']' : '[';
                }
                lut[a * 64 + b * 8 + c] = (uint8_t)(aa & 31);
            }
}

extern "C" int kasa_builtin_codon_table(uint8_t *lut366)
{
    if (!lut366) return fail(KASA_E_ARG, "kasa_builtin_codon_table: NULL argument");
    builtin_codon_table(lut366);
    return KASA_OK;
}

extern "C" int kasa_ctx_create(const kasa_index *ix, int kHigh, int kLow, int frames, const uint8_t *codonLut, kasa_ctx **out)
{
    if (!out) return fail(KASA_E_ARG, "kasa_ctx_create: out is NULL");
    *out = nullptr;
    if (!ix) return fail(KASA_E_ARG, "kasa_ctx_create: index is NULL");
    if (kHigh < kLow) std::swap(kHigh, kLow); // "-k <lower> <upper> is okay too" (README)
    if (kLow < 1 || kHigh > ix->letters()) return fail(KASA_E_ARG, "kasa_ctx_create: k range [%d,%d] outside [1,%d]", kLow, kHigh, ix->letters());
    if (frames != 1 && frames != 3 && frames != 6) return fail(KASA_E_ARG, "kasa_ctx_create: frames must be 1, 3 or 6");
    HIPCHK(hipSetDevice(ix->device));
    kasa_ctx *c = new (std::nothrow) kasa_ctx();
    if (!c) return fail(KASA_E_NOMEM, "host allocation failed");
    c->ix = ix; c->device = ix->device; c->kHigh = kHigh; c->kLow = kLow; c->nK = kHigh - kLow + 1; c->frames = frames;
    auto bail = [&](int code) { kasa_ctx_destroy(c); return code; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(KASA_E_HIP, "hipStreamCreate failed"));
    uint8_t lut[366];
    if (codonLut) memcpy(lut, codonLut, 366); else builtin_codon_table(lut);
    int rc = c->lut.reserve(512); if (rc) return bail(rc);
    if (hipMemcpy(c->lut.p, lut, 366, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(KASA_E_HIP, "LUT upload failed"));
    const size_t cells = (size_t)c->nK * ix->nTaxa * 8;
    if ((rc = c->cntUnique.reserve(cells)) || (rc = c->cntTotal.reserve(cells)) || (rc = c->cntAllHi.reserve(cells)) ||
        (rc = c->cntAllMid.reserve(cells)) || (rc = c->cntAllLo.reserve(cells)) || (rc = c->misc.reserve(512)))
        return bail(rc);
    *out = c;
    rc = kasa_profile_reset(c);
    if (rc) { *out = nullptr; return bail(rc); }
    return KASA_OK;
}

extern "C" int kasa_ctx_set_protein(kasa_ctx *c, int protein)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    c->protein = protein != 0;
    c->state = 0;          // an uploaded batch was measured with the other geometry
    return KASA_OK;
}

extern "C" void kasa_ctx_destroy(kasa_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device); // the index may already be gone: never touch it here
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    std::vector<DevBuf *> all = c->buffers();
    for (DevBuf *b : all) b->release();
    auto drop = [](StageTimer &t) {
        for (auto &pr : t.open) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (auto e : t.pool) (void)hipEventDestroy(e);
    };
    for (auto &t : c->timers) drop(t);
    for (auto &t : c->kernels) drop(t);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int kasa_profile_reset(kasa_ctx *c)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->ix->device));
    const size_t cells = (size_t)c->nK * c->ix->nTaxa * 8;
    HIPCHK(hipMemsetAsync(c->cntUnique.p, 0, cells, c->stream));
    HIPCHK(hipMemsetAsync(c->cntTotal.p, 0, cells, c->stream));
    HIPCHK(hipMemsetAsync(c->cntAllHi.p, 0, cells, c->stream));
    HIPCHK(hipMemsetAsync(c->cntAllMid.p, 0, cells, c->stream));
    HIPCHK(hipMemsetAsync(c->cntAllLo.p, 0, cells, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// upload + encode
// ------------------------------------------------------------------------------------------------
// Input geometry of one read (Read.hpp:633-654 padding, :1068-1078 marker, :36-57 k-mer count) for the three
// input kinds: mode 0 = DNA in 3 or 6 frames, 1 = DNA in one frame (--one), 2 = amino-acid input.
// body = padded read without the marker, L = body + marker, cnt = k-mers per strand.
enum { ENC_DNA = 0, ENC_ONE = 1, ENC_PROTEIN = 2 };
__host__ __device__ static inline void enc_geometry(int mode, int KL, int kLow, int64_t raw, int64_t &body, int64_t &L, int64_t &cnt)
{
    const int64_t K = KL;
    const int64_t marker = (mode == ENC_PROTEIN ? 1 : 3) * (K - (int64_t)kLow);
    body = raw;
    if (mode == ENC_PROTEIN) {
        if (body + marker < K) body = K - marker;
    } else if (body + marker < 3 * K) body = 3 * K - marker;   // --one: (len + m) / 3 < K  <=>  len + m < 3K
    L = body + marker;
    if (mode == ENC_PROTEIN) cnt = (L > K + 1) ? L - K + 1 : 0;
    else if (mode == ENC_ONE) { const int64_t t = L / 3; cnt = (t > K + 1) ? t - K + 1 : 0; }
    else cnt = (L > 3 * K + 1) ? L - 3 * K + 1 : 0;
}

// k-mers of every sequence (enc_geometry) and read, offsets relative to the batch, checks: what the encoder's tables are
// made of.  raw[] = the caller's offsets as uploaded.  err: bit 0 offsets not ascending, bit 1 bad read id; errAt: where.
__global__ void upload_geometry_kernel(const int64_t *__restrict__ raw, int64_t nSeq, const uint32_t *__restrict__ seqRead, int64_t nReads,
                                       int mode, int KL, int kLow, int strands, int64_t *__restrict__ baseOff, uint64_t *__restrict__ seqCnt,
                                       uint64_t *__restrict__ readCnt, uint32_t *__restrict__ err, unsigned long long *__restrict__ errAt,
                                       uint32_t *__restrict__ maxCnt)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nSeq) return;
    baseOff[s] = raw[s] - raw[0];
    if (s == nSeq) { seqCnt[s] = 0; return; }
    const int64_t len = raw[s + 1] - raw[s];
    const uint32_t r = seqRead ? seqRead[s] : (uint32_t)s;
    const uint32_t prev = s > 0 ? (seqRead ? seqRead[s - 1] : (uint32_t)(s - 1)) : 0u;
    if (len < 0) { atomicOr(err, 1u); atomicMin(errAt, (unsigned long long)s); seqCnt[s] = 0; return; }
    if ((int64_t)r >= nReads || r < prev) { atomicOr(err, 2u); atomicMin(errAt + 1, (unsigned long long)s); seqCnt[s] = 0; return; }
    int64_t body, L, perStrand = 0;
    if (len > 0) enc_geometry(mode, KL, kLow, len, body, L, perStrand);
    const uint64_t cnt = (uint64_t)perStrand * (uint64_t)strands;
    seqCnt[s] = cnt;
    if (seqRead) atomicAdd((unsigned long long *)&readCnt[r], (unsigned long long)cnt);   // the mates of a pair share a read
    else readCnt[r] = cnt;
}
__global__ void upload_max_kernel(const uint64_t *__restrict__ readCnt, int64_t nReads, uint32_t *__restrict__ maxCnt)
{
    uint32_t v = 0;                                               // (a few thousand wavefronts stride over the reads: one atomic each, not one per 64 reads)
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nReads; r += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t c = readCnt[r];
        v = max(v, (uint32_t)(c < 0xFFFFFFFFull ? c : 0xFFFFFFFFull));
    }
    for (int off = 32; off; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(maxCnt, v);
}

// nSeq sequences (offsets[nSeq+1]); seqRead[s] = read the sequence belongs to (ascending; NULL: sequence s is read s).
// Paired-end input hands both mates of a pair over as two sequences of one read (Read.hpp:834-1049): their k-mers
// carry the same read id, none spans the junction.  The host only copies: the tables (offsets relative to the batch,
// k-mers before every sequence and read) are made on the device -- the host loop over ten million reads took 0.1 s.
// resident: `bases` and `offsets` both lie in device memory and stay the caller's; the bases are read in place.
static int upload_impl(kasa_ctx *c, const uint8_t *bases, const int64_t *offsets, int64_t nSeq, const uint32_t *seqRead, int64_t nReads, bool resident = false)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (nSeq < 0 || nReads < 0 || (nSeq > 0 && (!offsets || !bases))) return fail(KASA_E_ARG, "kasa_batch_upload: bad arguments");
    if ((uint64_t)nReads >= 0xFFFFFFF0ull || (uint64_t)nSeq >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_batch_upload: more than 2^32 reads in one batch");
    HIPCHK(hipSetDevice(c->ix->device));
    c->state = 0; c->haveScores = false; c->grouped = false; c->slotOf = nullptr; c->payloadIsSlot = false; c->rankValid = false; c->txtValid = false; c->cohScores = nullptr; c->nReads = nReads; c->nSeq = nSeq; c->nQ = 0;
    c->recOut = nullptr; c->recSorted = false;
    const int64_t zero = 0;
    if (nSeq == 0) { offsets = &zero; resident = false; }
    int64_t ends[2] = {0, 0};                                               // offsets[0], offsets[nSeq]
    if (resident) {
        HIPCHK(hipMemcpyAsync(&ends[0], offsets, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&ends[1], offsets + nSeq, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    } else { ends[0] = offsets[0]; ends[1] = offsets[nSeq]; }
    if (ends[1] < ends[0]) return fail(KASA_E_ARG, "kasa_batch_upload: offsets are not ascending");
    const uint64_t nBases = (uint64_t)(ends[1] - ends[0]);
    int rc;
    if ((rc = resident ? KASA_OK : c->bases.reserve(nBases + 64)) || (rc = c->baseOff.reserve(((size_t)nSeq + 1) * 8)) ||
        (rc = c->kmerOff.reserve(((size_t)nReads + 1) * 8)) || (rc = c->seqOff.reserve(((size_t)nSeq + 1) * 8)) ||
        (rc = c->seqRead.reserve((size_t)nSeq * 4 + 64)) || (rc = c->rawOff.reserve(((size_t)nSeq + 1) * 8)))
        return rc;
    if (resident) c->basesPtr = bases + ends[0];
    else {
        if (nBases) HIPCHK(hipMemcpyAsync(c->bases.p, bases + ends[0], nBases, hipMemcpyDefault, c->stream));   // `bases` may live in device memory (a host that keeps its reads in HBM)
        c->basesPtr = c->bases.as<uint8_t>();
    }
    HIPCHK(hipMemcpyAsync(c->rawOff.p, offsets, ((size_t)nSeq + 1) * 8, resident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    c->haveSeqRead = seqRead != nullptr;
    if (seqRead && nSeq) HIPCHK(hipMemcpyAsync(c->seqRead.p, seqRead, (size_t)nSeq * 4, hipMemcpyHostToDevice, c->stream));
    // device: counts, checks, running sums
    uint32_t *flags = c->misc.as<uint32_t>() + 46;                      // [46] error bits, [47] most k-mers of a read
    unsigned long long *errAt = c->misc.as<unsigned long long>() + 24;  // [24], [25]: first offending sequence per error
    HIPCHK(hipMemsetAsync(flags, 0, 8, c->stream));
    HIPCHK(hipMemsetAsync(errAt, 0xFF, 16, c->stream));
    uint64_t *seqCnt = c->seqOff.as<uint64_t>(), *readCnt = c->kmerOff.as<uint64_t>();
    HIPCHK(hipMemsetAsync(readCnt, 0, ((size_t)nReads + 1) * 8, c->stream));
    upload_geometry_kernel<<<blocks_for((uint64_t)nSeq + 1, 256), 256, 0, c->stream>>>(c->rawOff.as<int64_t>(), nSeq, seqRead ? c->seqRead.as<uint32_t>() : nullptr, nReads,
        c->enc_mode(), c->K(), c->kLow, c->strands(), c->baseOff.as<int64_t>(), seqCnt, readCnt, flags, errAt, flags + 1);
    HIPCHK(hipGetLastError());
    if (nReads) upload_max_kernel<<<std::min<unsigned>(blocks_for((uint64_t)nReads, 256), 2048u), 256, 0, c->stream>>>(readCnt, nReads, flags + 1);
    size_t tmpBytes = 0, tmp2 = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, seqCnt, seqCnt, (uint64_t)0, (size_t)nSeq + 1, rocprim::plus<uint64_t>(), c->stream));
    HIPCHK(rocprim::exclusive_scan(nullptr, tmp2, readCnt, readCnt, (uint64_t)0, (size_t)nReads + 1, rocprim::plus<uint64_t>(), c->stream));
    if ((rc = c->sortTmp.reserve(std::max(tmpBytes, tmp2)))) return rc;
    HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, seqCnt, seqCnt, (uint64_t)0, (size_t)nSeq + 1, rocprim::plus<uint64_t>(), c->stream));
    HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmp2, readCnt, readCnt, (uint64_t)0, (size_t)nReads + 1, rocprim::plus<uint64_t>(), c->stream));
    uint32_t hFlags[2] = {0, 0};
    unsigned long long hAt[2] = {0, 0};
    uint64_t run = 0;
    HIPCHK(hipMemcpyAsync(hFlags, flags, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hAt, errAt, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&run, seqCnt + nSeq, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (hFlags[0] & 1u) return fail(KASA_E_ARG, "kasa_batch_upload: offsets are not ascending");
    if (hFlags[0] & 2u) {
        const int64_t sq = (int64_t)hAt[1];
        return fail(KASA_E_ARG, "kasa_batch_upload: sequence %lld names read %u (reads: %lld, ids must ascend)", (long long)sq, seqRead ? seqRead[sq] : (uint32_t)sq, (long long)nReads);
    }
    if (run >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_batch_upload: %llu k-mers exceed the 32-bit position range of one batch; split the batch", (unsigned long long)run);
    c->nQ = run; c->nEmitted = run; c->uniqueDone = false; c->readsUploaded = true; c->nBases = nBases; c->maxCnt = hFlags[1];
    c->state = 1;
    return KASA_OK;
}

extern "C" int kasa_batch_upload(kasa_ctx *c, const uint8_t *bases, const int64_t *offsets, int64_t nReads)
{
    KASA_GUARDED(upload_impl(c, bases, offsets, nReads, nullptr, nReads))
}

extern "C" int kasa_batch_upload_device(kasa_ctx *c, const uint8_t *basesDev, const int64_t *offsetsDev, int64_t nReads)
{
    KASA_GUARDED(upload_impl(c, basesDev, offsetsDev, nReads, nullptr, nReads, true))
}

extern "C" int kasa_batch_upload_segments(kasa_ctx *c, const uint8_t *bases, const int64_t *offsets, int64_t nSegments,
                                          const uint32_t *segmentRead, int64_t nReads)
{
    if (nSegments > 0 && !segmentRead) return fail(KASA_E_ARG, "kasa_batch_upload_segments: segmentRead is NULL");
    KASA_GUARDED(upload_impl(c, bases, offsets, nSegments, segmentRead, nReads))
}

// One wavefront per read.  The cleaned bases of a window chunk are staged in LDS as 3-bit codes, the
// codon letters are computed once per start position, each lane then packs K letters: window w starts at
// base w * ws and takes its letters at stride ls (DNA: ws 1, ls 3; --one: ws 3, ls 3, Read.hpp:223-261;
// amino-acid input: ws 1, ls 1 and the letters are the input itself, Read.hpp:60-81).
static constexpr int ENC_CHUNK = 512;                       // windows per chunk
static constexpr int ENC_WAVES = 4;
static constexpr int ENC_RANK_MAX = 512;                    // k-mers of a read the encoder ranks itself (payload = slot)
static constexpr uint32_t ENC_LONG_MIN = 65536;             // k-mers from which a SEQUENCE is encoded by all wavefronts, a chunk each
__global__ void enc_long_list_kernel(const uint64_t *__restrict__ seqOff, int64_t nSeq, uint64_t longMin, uint32_t *__restrict__ list, uint32_t *__restrict__ n)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nSeq && seqOff[r + 1] - seqOff[r] >= longMin) list[atomicAdd(n, 1u)] = (uint32_t)r;
}

#define LDS_WAVE_SYNC_ENC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
// RM: capacity of the per-read ranking (k-mers of a read): 512, or 192 when no read of the batch has more (150-bp reads in
// three frames have 130) -- a third of the LDS and three ranking rounds instead of eight, so more wavefronts are resident.
template <class Key, int RM = ENC_RANK_MAX, bool LONGK = false>        // LONGK: the second launch over the listed long sequences
__global__ __launch_bounds__(64 * ENC_WAVES, RM <= 192 ? 6 : 1) void encode_kernel(
    const uint8_t *__restrict__ bases, const int64_t *__restrict__ baseOff, const uint64_t *__restrict__ kmerOff,
    const uint32_t *__restrict__ seqRead, int64_t nReads, int kLow, int strands, int mode, const uint8_t *__restrict__ lutG,
    Key *__restrict__ outKmer, uint32_t *__restrict__ outRead, int rankSlots,
    const uint32_t *__restrict__ longList, const uint32_t *__restrict__ nLongPtr, uint64_t longMin)
{
    // A sequence of longMin k-mers and more (a contig: one wavefront took 87 ms for 9.6 M k-mers) is not one wavefront's: the
    // first launch skips it, a second one (longList: the listed sequences) deals ITS chunks out to all wavefronts.
    constexpr int KLETTERS = KeyTraits<Key>::LETTERS;
    constexpr int ENC_SPAN = ENC_CHUNK + 3 * KLETTERS;      // bases needed for one chunk (+ slack)
    __shared__ uint8_t sLut[384];
    __shared__ uint8_t sCode[ENC_WAVES][ENC_SPAN + 8];
    __shared__ uint8_t sLetter[ENC_WAVES][ENC_SPAN + 8];
    // rankSlots: the payload of a k-mer is its SLOT -- the read's first slot (= its first k-mer's index) plus the rank of the
    // k-mer among the read's k-mers (ties in window order, as the stable sort would leave them) -- so that after the sort
    // every query knows its place in the read-major, sorted-within-read record array (group_kernel writes there).  All
    // k-mers of a read (both strands) wait in LDS for that; the host enables it when every read has at most ENC_RANK_MAX.
    __shared__ Key sKey[ENC_WAVES][RM];
    __shared__ uint32_t sTaken[ENC_WAVES][128 + RM / 2];              // bucket counts, bucket starts, members (halfwords)
    for (int i = threadIdx.x; i < 366; i += blockDim.x) sLut[i] = lutG[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t wavesTotal = (int64_t)gridDim.x * ENC_WAVES;
    const int ws = (mode == ENC_ONE) ? 3 : 1;
    const int ls = (mode == ENC_PROTEIN) ? 1 : 3;
    const int tail = (KLETTERS - 1) * ls + ((mode == ENC_PROTEIN) ? 1 : 3);   // bases of the last window
    const int chunk = (mode == ENC_ONE) ? ENC_CHUNK / 3 : ENC_CHUNK;
    constexpr bool longK = LONGK;
    const int64_t waveId = (int64_t)blockIdx.x * ENC_WAVES + wv;
    const int64_t nWork = longK ? (int64_t)*nLongPtr : nReads;
    for (int64_t wi = longK ? 0 : waveId; wi < nWork; wi += longK ? 1 : wavesTotal) {
        const int64_t r = longK ? (int64_t)longList[wi] : wi;
        const int64_t b0 = baseOff[r];
        const int64_t raw = baseOff[r + 1] - b0;
        if (raw <= 0) continue;
        int64_t body, L, cnt;
        enc_geometry(mode, KLETTERS, kLow, raw, body, L, cnt);
        if (cnt == 0) continue;
        const uint64_t o0 = kmerOff[r];
        if (!longK && longMin != ~0ull && kmerOff[r + 1] - o0 >= longMin) continue;   // (the second launch's)
        const uint32_t rid = seqRead ? seqRead[r] : (uint32_t)r;          // paired-end: both mates carry the pair's id
        for (int s = 0; s < strands; ++s) {
            for (int64_t w0 = longK ? waveId * chunk : 0; w0 < cnt; w0 += longK ? wavesTotal * chunk : chunk) {
                const int nw = (int)((cnt - w0 < chunk) ? cnt - w0 : chunk);
                const int span = (nw - 1) * ws + tail;      // bases w0*ws .. w0*ws+span-1
                const int64_t base0 = w0 * ws;
                if (mode == ENC_PROTEIN) {
                    for (int i = lane; i < span; i += 64) {
                        const int64_t pos = base0 + i;
                        uint8_t ch = (pos < raw) ? bases[b0 + pos] : (uint8_t)'^';   // padding and marker are '^'
                        if (ch == '*') ch = '[';                                    // Read.hpp:663-667
                        sLetter[wv][i] = ch & 31;
                    }
                } else {
                    for (int i = lane; i < span; i += 64) {
                        const int64_t pos = base0 + i;
                        uint8_t code;
                        if (pos >= body) code = 4;               // X: padding and marker
                        else {
                            const int64_t src = (s == 0) ? pos : (body - 1 - pos);
                            if (src >= raw) code = 4;            // padding X (reverse strand sees it first)
                            else {
                                const uint8_t ch = bases[b0 + src];
                                const uint8_t up = ch & 0xDF;
                                const bool ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
                                code = ok ? (uint8_t)((ch & 14) >> 1) : (uint8_t)5;  // everything else is Z
                                if (s == 1 && code < 4) code ^= 2;                   // A<->T, C<->G
                            }
                        }
                        sCode[wv][i] = code;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // LDS writes of this wave are done
                    __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < span - 2; i += 64)
                        sLetter[wv][i] = sLut[sCode[wv][i] * 64 + sCode[wv][i + 1] * 8 + sCode[wv][i + 2]];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < nw; i += 64) {
                    // (six letters at a time in 32 bits -- one shift-or each -- then one wide shift per group)
                    Key v = 0;
                    const uint8_t *lt = &sLetter[wv][i * ws];
#pragma unroll
                    for (int j0 = 0; j0 < KLETTERS; j0 += 6) {
                        uint32_t part = 0;
#pragma unroll
                        for (int j = j0; j < j0 + 6 && j < KLETTERS; ++j) part = (part << 5) | lt[j * ls];
                        const int got = KLETTERS - j0 < 6 ? KLETTERS - j0 : 6;
                        v = (v << (5 * got)) | (Key)part;
                    }
                    const uint64_t o = o0 + (uint64_t)s * cnt + w0 + i;
                    outKmer[o] = v;
                    if (rankSlots) sKey[wv][s * (int)cnt + (int)w0 + i] = v << KeyTraits<Key>::SHIFT;   // (--one: several chunks per strand; strands * cnt <= ENC_RANK_MAX)
                    else outRead[o] = rid;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (rankSlots) {
            // rank = k-mers of the read that are smaller (equal ones: those with a lower index, as the stable sort leaves
            // them).  Comparing every k-mer with every other took 130 steps per k-mer; instead the k-mers are dealt into 64
            // buckets by their top six bits (lane b owns bucket b: counts by LDS atomics, starts by one running sum over the
            // lanes) and a k-mer is compared with the members of its own bucket only -- two or three on average, a dozen for
            // the most common first letter.
            const int n = strands * (int)cnt;
            constexpr int PER = RM / 64;
            constexpr int TOPSH = 8 * (int)sizeof(Key) - 6;
            uint32_t *sCount = &sTaken[wv][0], *sStart = &sTaken[wv][64];
            uint16_t *sMember = reinterpret_cast<uint16_t *>(&sTaken[wv][128]);   // item indices grouped by bucket: n <= RM halfwords
            sCount[lane] = 0u;
            LDS_WAVE_SYNC_ENC();
            uint32_t place[PER];                                         // arrival number inside the bucket
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = lane + 64 * q;
                place[q] = i < n ? atomicAdd(&sCount[(uint32_t)(sKey[wv][i] >> TOPSH)], 1u) : 0u;
            }
            LDS_WAVE_SYNC_ENC();
            const uint32_t mineCount = sCount[lane];
            sStart[lane] = wave_incl_sum(mineCount) - mineCount;
            LDS_WAVE_SYNC_ENC();
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = lane + 64 * q;
                if (i < n) sMember[sStart[(uint32_t)(sKey[wv][i] >> TOPSH)] + place[q]] = (uint16_t)i;
            }
            LDS_WAVE_SYNC_ENC();
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = lane + 64 * q;
                if (i >= n) continue;
                const Key me = sKey[wv][i];
                const uint32_t b = (uint32_t)(me >> TOPSH), first = sStart[b], members = sCount[b];
                uint32_t rk = first;
                for (uint32_t m = 0; m < members; ++m) {
                    const int other = (int)sMember[first + m];
                    const Key ko = sKey[wv][other];
                    rk += (ko < me || (ko == me && other < i)) ? 1u : 0u;
                }
                outRead[o0 + (uint64_t)i] = (uint32_t)(o0 + rk);
            }
            LDS_WAVE_SYNC_ENC();
        }
    }
}
#undef LDS_WAVE_SYNC_ENC

extern "C" int kasa_batch_encode(kasa_ctx *c, uint64_t *nKmers)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (c->state < 1) return fail(KASA_E_STATE, "kasa_batch_encode: no batch uploaded");
    HIPCHK(hipSetDevice(c->ix->device));
    int rc;
    const uint64_t nQ = c->nQ;
    if ((rc = c->qKmerA.reserve(nQ * c->keyBytes() + 64)) || (rc = c->qReadA.reserve(nQ * 4 + 64))) return rc;
    hipEvent_t a, b;
    if ((rc = timer_begin(c, c->timers[KASA_STAGE_ENCODE], &a, &b))) return rc;
    // the encoder ranks a read's k-mers itself (payload of the sort = slot) when every read is one sequence with few enough
    // k-mers; else the payload is the read id and the slots come from a stable sort by read (slots_from_reads)
    const int rankSlots = (!c->haveSeqRead && c->maxCnt <= (uint32_t)ENC_RANK_MAX && !(c->debugFlags & 8)) ? 1 : 0;
    c->payloadIsSlot = rankSlots != 0;
    const uint64_t longMin = (!rankSlots && c->maxCnt >= ENC_LONG_MIN) ? (uint64_t)ENC_LONG_MIN : ~0ull;   // (maxCnt: k-mers of the batch's longest READ)
    if (c->nSeq > 0 && nQ > 0) {
        const unsigned blocks = (unsigned)std::min<int64_t>((c->nSeq + ENC_WAVES - 1) / ENC_WAVES, 256 * 16);
        if (c->ix->wide && rankSlots && c->maxCnt <= 192u)
            encode_kernel<key128, 192><<<blocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(),
                c->seqOff.as<uint64_t>(), c->haveSeqRead ? c->seqRead.as<uint32_t>() : nullptr, c->nSeq, c->kLow, c->strands(), c->enc_mode(),
                c->lut.as<uint8_t>(), c->qKmerA.as<key128>(), c->qReadA.as<uint32_t>(), rankSlots, nullptr, nullptr, longMin);
        else if (c->ix->wide)
            encode_kernel<key128><<<blocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(),
                c->seqOff.as<uint64_t>(), c->haveSeqRead ? c->seqRead.as<uint32_t>() : nullptr, c->nSeq, c->kLow, c->strands(), c->enc_mode(),
                c->lut.as<uint8_t>(), c->qKmerA.as<key128>(), c->qReadA.as<uint32_t>(), rankSlots, nullptr, nullptr, longMin);
        else if (rankSlots && c->maxCnt <= 192u)
            encode_kernel<uint64_t, 192><<<blocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(),
                c->seqOff.as<uint64_t>(), c->haveSeqRead ? c->seqRead.as<uint32_t>() : nullptr, c->nSeq, c->kLow, c->strands(), c->enc_mode(),
                c->lut.as<uint8_t>(), c->qKmerA.as<uint64_t>(), c->qReadA.as<uint32_t>(), rankSlots, nullptr, nullptr, longMin);
        else
            encode_kernel<uint64_t><<<blocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(),
                c->seqOff.as<uint64_t>(), c->haveSeqRead ? c->seqRead.as<uint32_t>() : nullptr, c->nSeq, c->kLow, c->strands(), c->enc_mode(),
                c->lut.as<uint8_t>(), c->qKmerA.as<uint64_t>(), c->qReadA.as<uint32_t>(), rankSlots, nullptr, nullptr, longMin);
        HIPCHK(hipGetLastError());
        if (longMin != ~0ull) {                                            // the long sequences: their chunks over all wavefronts
            if ((rc = c->encLong.reserve(((size_t)c->nSeq + 1) * 4 + 64))) return rc;
            uint32_t *nLong = c->encLong.as<uint32_t>(), *list = nLong + 1;
            HIPCHK(hipMemsetAsync(nLong, 0, 4, c->stream));
            enc_long_list_kernel<<<blocks_for((uint64_t)c->nSeq, 256), 256, 0, c->stream>>>(c->seqOff.as<uint64_t>(), c->nSeq, longMin, list, nLong);
            const unsigned lblocks = 256 * 8;
            if (c->ix->wide)
                encode_kernel<key128, ENC_RANK_MAX, true><<<lblocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(),
                    c->seqOff.as<uint64_t>(), c->haveSeqRead ? c->seqRead.as<uint32_t>() : nullptr, c->nSeq, c->kLow, c->strands(), c->enc_mode(),
                    c->lut.as<uint8_t>(), c->qKmerA.as<key128>(), c->qReadA.as<uint32_t>(), 0, list, nLong, longMin);
            else
                encode_kernel<uint64_t, ENC_RANK_MAX, true><<<lblocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(),
                    c->seqOff.as<uint64_t>(), c->haveSeqRead ? c->seqRead.as<uint32_t>() : nullptr, c->nSeq, c->kLow, c->strands(), c->enc_mode(),
                    c->lut.as<uint8_t>(), c->qKmerA.as<uint64_t>(), c->qReadA.as<uint32_t>(), 0, list, nLong, longMin);
            HIPCHK(hipGetLastError());
        }
    }
    if ((rc = timer_end(c, c->timers[KASA_STAGE_ENCODE], a, b))) return rc;
    c->qKmer = c->qKmerA.p;
    c->qRead = c->qReadA.as<uint32_t>();
    c->state = 2;
    if (nKmers) *nKmers = nQ;
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// sort + lookup
// ------------------------------------------------------------------------------------------------
// One query per thread, TILE queries per workgroup (the same tiling as group_kernel).  Level 1 of the
// prefix table narrows the search to the entries sharing the top `tb` key bits, a binary search
// finishes it.  Consecutive threads hold consecutive sorted queries, so table and index reads of a
// wavefront fall into a few cache lines.  Also emits, per tile and level, the first position that
// closes a group ("special"), the seed of the flush-order computation.
template <class Key>
__global__ __launch_bounds__(TILE_THREADS) void lookup_kernel(
    const Key *__restrict__ qKmer, uint32_t nQ, const Key *__restrict__ idxKmer, uint32_t nIdx,
    const uint32_t *__restrict__ table, int tb, int kHigh, int kLow, uint8_t *__restrict__ depth,
    uint32_t *__restrict__ rep, uint32_t *__restrict__ tileFirst, uint32_t nTiles)
{
    __shared__ uint32_t sFirst[MAX_LEVELS];
    const int nK = kHigh - kLow + 1;
    if (threadIdx.x < MAX_LEVELS) sFirst[threadIdx.x] = NOPOS;
    __syncthreads();
    const uint32_t base = blockIdx.x * TILE;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t p = base + it * TILE_THREADS + threadIdx.x;   // striped: coalesced loads
        if (p >= nQ) continue;
        const Key q = qKmer[p];
        const uint64_t bkt = (uint64_t)(q >> (KeyTraits<Key>::BITS - tb));
        uint32_t lo = bkt ? table[bkt - 1] : 0u;
        uint32_t hi = table[bkt];
        while (lo < hi) {                                           // first entry >= q
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (idxKmer[mid] < q) lo = mid + 1; else hi = mid;
        }
        const int la = (lo < nIdx) ? lcp_letters<Key>(q, idxKmer[lo]) : 0;
        const int lb = (lo > 0) ? lcp_letters<Key>(q, idxKmer[lo - 1]) : 0;
        int L = la >= lb ? la : lb;
        const uint32_t r = la >= lb ? lo : lo - 1;
        int d = 0;
        if (L >= RANGE_LETTERS) {                                   // the 6-letter prefix exists (Trie.hpp:494)
            if (L > kHigh) L = kHigh;
            d = L;
            for (int k = kLow; k <= L; ++k)                         // '^' ends the query (Compare.hpp:836,897)
                if (((uint32_t)(q >> (5 * (KeyTraits<Key>::LETTERS - k))) & 31u) == 30u) { d = k - 1; break; }
            if (d < kLow) d = 0;
        }
        depth[p] = (uint8_t)d;
        rep[p] = r;
        const int ql = (p == 0) ? 0 : lcp_letters<Key>(qKmer[p - 1], q);
        for (int lv = 0; lv < nK; ++lv) {
            const int k = kHigh - lv;
            const bool special = (ql < RANGE_LETTERS) || (ql < group_letters(k) && d >= k);
            // p grows with the lane: the lowest special lane holds this wavefront's minimum
            const unsigned long long m = __ballot(special);
            if (special && (m & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicMin(&sFirst[lv], p);
        }
    }
    __syncthreads();
    if (threadIdx.x < nK) tileFirst[(size_t)threadIdx.x * nTiles + blockIdx.x] = sFirst[threadIdx.x];
}

// ---- streaming variant --------------------------------------------------------------------------
// When a batch is dense (the normal case: ~3 queries per index record at 10 M reads vs 4e8 records), a
// tile of 1024 consecutive sorted queries touches a short contiguous span of the index.  tile_bounds
// finds that span once per tile (prefix table + binary search, one thread per tile edge); lookup_tile
// then streams the span into LDS with coalesced loads and answers all 1024 queries from LDS.  Every
// index record and every query is read from HBM once: the merge-join lower bound of SURVEY.md 8(d).
static constexpr int LSPAN = 1536;   // index records staged per tile (12 KiB); larger spans use lookup_kernel's path

template <class Key>
__device__ __forceinline__ uint32_t lower_bound_global(const Key *__restrict__ idxKmer, const uint32_t *__restrict__ table,
                                                       int tb, Key q)
{
    const uint64_t bkt = (uint64_t)(q >> (KeyTraits<Key>::BITS - tb));
    uint32_t lo = bkt ? table[bkt - 1] : 0u;
    uint32_t hi = table[bkt];
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (idxKmer[mid] < q) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <class Key>
__global__ void tile_bounds_kernel(const Key *__restrict__ qKmer, uint32_t nQ, const Key *__restrict__ idxKmer,
                                   const uint32_t *__restrict__ table, int tb, uint32_t nTiles, uint32_t *__restrict__ bounds)
{
    // edge 2t: start of the level-1 bucket of the tile's first query; edge 2t+1: end of the bucket of its last query.
    // The true lower bounds of all the tile's queries lie inside [bounds[2t], bounds[2t+1]] -- two table reads, no search.
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * nTiles) return;
    const uint32_t t = i >> 1;
    uint32_t p = t * TILE + ((i & 1) ? (TILE - 1) : 0);
    if (p >= nQ) p = nQ - 1;
    const uint64_t bkt = (uint64_t)(qKmer[p] >> (KeyTraits<Key>::BITS - tb));
    bounds[i] = (i & 1) ? table[bkt] : (bkt ? table[bkt - 1] : 0u);
    (void)idxKmer;
}

// One workgroup per tile, a thread owns ITEMS = 4 CONSECUTIVE sorted queries (32 bytes, two 16-byte loads; a
// wavefront reads 2 KiB contiguous), so a query's predecessor is in a register, results leave as one 16-byte and
// one 4-byte store, and the per-level "first special position" needs one ballot per level.
template <class Key>
__global__ __launch_bounds__(TILE_THREADS) void lookup_tile_kernel(
    const Key *__restrict__ qKmer, uint32_t nQ, const Key *__restrict__ idxKmer, uint32_t nIdx,
    const uint32_t *__restrict__ table, int tb, const uint32_t *__restrict__ bounds, int kHigh, int kLow,
    uint8_t *__restrict__ depth, uint32_t *__restrict__ rep, uint32_t *__restrict__ tileFirst, uint32_t nTiles)
{
    constexpr int KLETTERS = KeyTraits<Key>::LETTERS;
    constexpr Key HAT_ALL = field_repeat<Key>(30), LOW4_ALL = field_repeat<Key>(15), BIT4_ALL = field_repeat<Key>(16);
    __shared__ Key sIdx[LSPAN + 2];
    __shared__ uint32_t sFirst[MAX_LEVELS];
    const int nK = kHigh - kLow + 1;
    const int tid = threadIdx.x;
    if (tid < MAX_LEVELS) sFirst[tid] = NOPOS;
    const uint32_t base = blockIdx.x * TILE;
    const uint32_t p0 = base + ITEMS * tid;
    const bool fullTile = base + TILE <= nQ;
    Key qv[ITEMS];
    if constexpr (sizeof(Key) == 8) {
        if (fullTile) {
            const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(qKmer + p0);
            const ulonglong2 v1 = *reinterpret_cast<const ulonglong2 *>(qKmer + p0 + 2);
            qv[0] = v0.x; qv[1] = v0.y; qv[2] = v1.x; qv[3] = v1.y;
        } else {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) qv[it] = (p0 + it < nQ) ? qKmer[p0 + it] : 0ull;
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) qv[it] = (p0 + it < nQ) ? qKmer[p0 + it] : (Key)0;   // 16-byte loads as they are
    }
    const Key qBefore = (p0 > 0 && p0 < nQ) ? qKmer[p0 - 1] : (Key)0;
    const uint2 bd = *reinterpret_cast<const uint2 *>(bounds + 2 * (size_t)blockIdx.x);
    const uint32_t ilo = bd.x ? bd.x - 1 : 0u;                         // predecessor of the first query
    const uint32_t ihi = (bd.y < nIdx) ? bd.y + 1 : nIdx;              // successor of the last query
    const uint32_t span = ihi - ilo;
    const bool staged = span <= (uint32_t)LSPAN;
    if (staged) {
        if constexpr (sizeof(Key) == 8) {
            const uint32_t a0 = ilo & ~1u;                             // 16-byte aligned start (one extra record at most)
            const uint32_t shift = ilo - a0;
            for (uint32_t i = 2 * tid; i < span + shift; i += 2 * TILE_THREADS) {
                if (a0 + i + 1 < nIdx) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(idxKmer + a0 + i);
                    if (i >= shift) sIdx[i - shift] = v.x;
                    if (i + 1 >= shift && i + 1 - shift < span) sIdx[i + 1 - shift] = v.y;
                } else if (a0 + i < nIdx && i >= shift) sIdx[i - shift] = idxKmer[a0 + i];
            }
        } else {
            for (uint32_t i = tid; i < span; i += TILE_THREADS) sIdx[i] = idxKmer[ilo + i];
        }
    }
    __syncthreads();
    uint32_t pos[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) pos[it] = 0;
    if (staged) {
        // branch-free lower bound for the thread's first query; its next three queries are consecutive in sorted
        // order, so their lower bounds lie a step or two further on: gallop from the previous answer
        uint32_t step = 1;
        while ((step << 1) <= span) step <<= 1;
        uint32_t a = 0;
        for (; step > 0; step >>= 1) {
            const uint32_t probe = a + step;
            if (probe <= span && sIdx[probe - 1] < qv[0]) a = probe;
        }
        pos[0] = a;
#pragma unroll
        for (int it = 1; it < ITEMS; ++it) {
            while (a < span && sIdx[a] < qv[it]) ++a;
            pos[it] = a;
        }
    }
    uint32_t outRep[ITEMS];
    uint32_t outD[ITEMS];
    uint32_t special[ITEMS];                                           // bit lv: the position opens a level-(kHigh-lv) group
    const uint32_t allLv = (nK >= 32) ? 0xFFFFFFFFu : ((1u << nK) - 1u);   // nK <= MAX_LEVELS
    const Key hatLetters = ((Key)2 << (5 * (KLETTERS - kLow) + 4)) - (Key)1;   // the bits of letters kLow..K
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t p = p0 + it;
        outRep[it] = 0; outD[it] = 0; special[it] = 0;
        if (p >= nQ) continue;
        const Key q = qv[it];
        uint32_t lo;
        Key eLo = 0, ePrev = 0;
        bool hasLo, hasPrev;
        if (staged) {
            const uint32_t a = pos[it];
            lo = ilo + a;
            hasLo = a < span; hasPrev = a > 0;
            if (hasLo) eLo = sIdx[a];
            if (hasPrev) ePrev = sIdx[a - 1];
            if (!hasPrev && lo > 0) { ePrev = idxKmer[lo - 1]; hasPrev = true; }       // only at the very ends of the index
            if (!hasLo && lo < nIdx) { eLo = idxKmer[lo]; hasLo = true; }
        } else {
            lo = lower_bound_global<Key>(idxKmer, table, tb, q);
            hasLo = lo < nIdx; hasPrev = lo > 0;
            if (hasLo) eLo = idxKmer[lo];
            if (hasPrev) ePrev = idxKmer[lo - 1];
        }
        // letters shared with the closer neighbour: the smaller XOR has the longer common prefix
        const Key xa = hasLo ? (q ^ eLo) : ~(Key)0;
        const Key xb = hasPrev ? (q ^ ePrev) : ~(Key)0;
        const int la = lcp_of_xor<Key>(xa);
        int L = lcp_of_xor<Key>(xa < xb ? xa : xb);
        const uint32_t r = (la == L) ? lo : lo - 1;                     // ties go to the lower bound itself
        int d = 0;
        if (L >= RANGE_LETTERS) {                                       // the 6-letter prefix exists (Trie.hpp:494)
            d = L > kHigh ? kHigh : L;
            // '^' ends the query (Compare.hpp:836,897): first letter >= kLow that equals 30, found without a loop --
            // per 5-bit field, bit 4 of ((x & 01111b) + 01111b) | x is set iff the field of x = q ^ "^^^..." is non-zero
            const Key x = q ^ HAT_ALL;
            const Key z = ~(((x & LOW4_ALL) + LOW4_ALL) | x) & BIT4_ALL & hatLetters;
            if (z) {
                const int khat = KLETTERS - ((int)(8 * sizeof(Key)) - 5 - clz_key(z)) / 5;   // bit 5(K-k)+4 belongs to letter k
                if (khat <= d) d = khat - 1;
            }
            if (d < kLow) d = 0;
        }
        outD[it] = (uint32_t)d;
        outRep[it] = r;
        const Key prevQ = it ? qv[it - 1] : qBefore;
        const int ql = (p == 0) ? 0 : lcp_letters<Key>(prevQ, q);
        // special for level k: a new range (ql < 6), or a new matched group: ql < k <= d
        uint32_t m = allLv;
        if (ql >= RANGE_LETTERS) m = (d > ql) ? ((((2u << (d - ql - 1)) - 1u) << (kHigh - d)) & allLv) : 0u;
        special[it] = m;
    }
    // positions grow with the lane and with the item: per level the lowest lane holding a special position has the
    // wavefront's minimum
    const uint32_t anySpecial = special[0] | special[1] | special[2] | special[3];
    for (int lv = 0; lv < nK; ++lv) {
        const bool has = (anySpecial >> lv) & 1u;
        const unsigned long long mk = __ballot(has);
        if (has && (mk & ((1ull << (tid & 63)) - 1ull)) == 0ull) {
            const uint32_t it = ((special[0] >> lv) & 1u) ? 0u : ((special[1] >> lv) & 1u) ? 1u : ((special[2] >> lv) & 1u) ? 2u : 3u;
            atomicMin(&sFirst[lv], p0 + it);
        }
    }
    if (fullTile) {
        *reinterpret_cast<uint4 *>(rep + p0) = make_uint4(outRep[0], outRep[1], outRep[2], outRep[3]);
        *reinterpret_cast<uchar4 *>(depth + p0) = make_uchar4((uint8_t)outD[0], (uint8_t)outD[1], (uint8_t)outD[2], (uint8_t)outD[3]);
    } else {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
            if (p0 + it < nQ) { rep[p0 + it] = outRep[it]; depth[p0 + it] = (uint8_t)outD[it]; }
    }
    __syncthreads();
    if (tid < nK) tileFirst[(size_t)tid * nTiles + blockIdx.x] = sFirst[tid];
}

// tileNext[lv][t] = first special position in any tile after t (or nQ): a suffix minimum over the tiles of a level, in
// three small steps -- inside chunks of 1024 tiles (one workgroup each), over the chunk minima (one workgroup per level,
// serial over its chunks), and the two combined.  (One workgroup per level walking all 1.3 M tiles of a 10 M-read batch
// took 2.2 ms.)
__device__ __forceinline__ void suffix_min_1024(uint32_t *sh, uint32_t v, uint32_t &excl, uint32_t &all)
{
    // threads hold the values in REVERSED order (thread 0 = last element): an inclusive min-scan, then shifted by one
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t o = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : NOPOS;
        __syncthreads();
        if (o < sh[threadIdx.x]) sh[threadIdx.x] = o;
        __syncthreads();
    }
    excl = (threadIdx.x == 0) ? NOPOS : sh[threadIdx.x - 1];
    all = sh[1023];
    __syncthreads();
}

// grid (chunks, levels): out = minimum over the later tiles of the same chunk (NOPOS: none), chunkMin = the chunk's minimum
__global__ __launch_bounds__(1024) void tile_suffix_local_kernel(const uint32_t *__restrict__ tileFirst, uint32_t *__restrict__ tileNext,
                                                                 uint32_t *__restrict__ chunkMin, uint32_t nTiles, uint32_t nChunks)
{
    __shared__ uint32_t sh[1024];
    const uint32_t lv = blockIdx.y, chunk = blockIdx.x;
    const uint64_t end = std::min<uint64_t>((uint64_t)(chunk + 1) * 1024u, nTiles);
    const int64_t t = (int64_t)end - 1 - threadIdx.x;                 // thread 0 takes the last tile of the chunk
    const bool in = t >= (int64_t)chunk * 1024;
    uint32_t excl, all;
    suffix_min_1024(sh, in ? tileFirst[(size_t)lv * nTiles + t] : NOPOS, excl, all);
    if (in) tileNext[(size_t)lv * nTiles + t] = excl;
    if (threadIdx.x == 0) chunkMin[(size_t)lv * nChunks + chunk] = all;
}

// one workgroup per level: chunkNext[c] = minimum over the chunks after c, or `last` (the number of queries)
__global__ __launch_bounds__(1024) void tile_suffix_kernel(const uint32_t *__restrict__ in0, uint32_t *__restrict__ out0,
                                                           uint32_t n, uint32_t last)
{
    __shared__ uint32_t sh[1024];
    const int lv = blockIdx.x;
    const uint32_t *in = in0 + (size_t)lv * n;
    uint32_t *out = out0 + (size_t)lv * n;
    uint32_t carry = last;
    for (int64_t hiT = (int64_t)n; hiT > 0; hiT -= 1024) {
        const int64_t t = hiT - 1 - threadIdx.x;
        uint32_t excl, all;
        suffix_min_1024(sh, (t >= 0) ? in[t] : NOPOS, excl, all);
        if (t >= 0) out[t] = excl < carry ? excl : carry;
        if (all < carry) carry = all;
    }
}

__global__ void tile_suffix_apply_kernel(uint32_t *__restrict__ tileNext, const uint32_t *__restrict__ chunkNext, uint32_t nTiles, uint32_t nChunks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lv = blockIdx.y;
    if (t >= nTiles) return;
    const size_t at = (size_t)lv * nTiles + t;
    if (tileNext[at] == NOPOS) tileNext[at] = chunkNext[(size_t)lv * nChunks + (t >> 10)];   // positions ascend with the tile: a later chunk never undercuts this one
}

// tileFirst -> tileNext for all levels
static int tile_suffix(kasa_ctx *c, uint32_t nTiles)
{
    if (nTiles == 0) return KASA_OK;
    const uint32_t nChunks = (nTiles + 1023u) / 1024u;
    int rc = c->tileChunks.reserve((size_t)2 * c->nK * nChunks * 4 + 64);
    if (rc) return rc;
    uint32_t *chunkMin = c->tileChunks.as<uint32_t>(), *chunkNext = chunkMin + (size_t)c->nK * nChunks;
    tile_suffix_local_kernel<<<dim3(nChunks, (unsigned)c->nK), 1024, 0, c->stream>>>(c->tileFirst.as<uint32_t>(), c->tileNext.as<uint32_t>(), chunkMin, nTiles, nChunks);
    tile_suffix_kernel<<<c->nK, 1024, 0, c->stream>>>(chunkMin, chunkNext, nChunks, (uint32_t)c->nQ);
    tile_suffix_apply_kernel<<<dim3(blocks_for(nTiles, 256), (unsigned)c->nK), 256, 0, c->stream>>>(c->tileNext.as<uint32_t>(), chunkNext, nTiles, nChunks);
    HIPCHK(hipGetLastError());
    return KASA_OK;
}

template <class Key>
__global__ void unique_flag_kernel(const Key *__restrict__ kmer, const uint32_t *__restrict__ read, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || kmer[i] != kmer[i - 1] || read[i] != read[i - 1]) ? 1u : 0u;
}

__global__ void read_count_kernel(const uint32_t *__restrict__ read, uint32_t n, uint64_t *__restrict__ cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd((unsigned long long *)&cnt[read[i]], 1ull);
}

template <class Key>
__global__ void unique_scatter_kernel(const Key *__restrict__ kmer, const uint32_t *__restrict__ read, uint32_t n,
                                      const uint32_t *__restrict__ slot, Key *__restrict__ outKmer, uint32_t *__restrict__ outRead)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = slot[i];
    if (i == 0 || sl != slot[i - 1]) { outKmer[sl - 1] = kmer[i]; outRead[sl - 1] = read[i]; }
}

// read of a slot: slots of read r are kmerOff[r] .. kmerOff[r+1]
__global__ void slot_to_read_kernel(const uint32_t *__restrict__ slot, uint32_t n, const uint64_t *__restrict__ kmerOff, uint32_t nReads,
                                    uint32_t *__restrict__ read)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = slot[i];
    uint32_t lo = 0, hi = nReads;                                  // last r with kmerOff[r] <= s
    while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (kmerOff[mid] <= s) lo = mid; else hi = mid; }
    read[i] = lo;
}

template <class Key> static int lookup_part(kasa_ctx *c);

// Second half of the query sort.  The pairs arrive ordered (stably) by the top SORT_TOP bits of the key; queries that
// share those bits -- a *bucket*: six letters and two bits of the seventh, i.e. the copies of a k-mer the reads' coverage brings
// and the few other k-mers with that prefix, 17 members for the average query of the bench data (three with SORT_TOP_OLD) --
// are contiguous.  Every query finds its place inside its bucket by counting: the bucket members before it with a key
// not larger, those after it with a smaller key (= the stable order).  One pass over the pairs instead of the remaining
// 28 (93) bits' worth of radix passes (round 4: a fifth radix pass costs 8 ms at 1.15e9 pairs, the longer buckets 4).  Buckets of more than SORT_BUCKET_LIMIT members (an input that repeats itself) are
// copied through and listed in `bigHead`; the caller sorts each of them by its remaining bits.
static constexpr unsigned SORT_TOP = 32, SORT_TOP_OLD = 40;
static constexpr uint32_t SORT_BUCKET_LIMIT = 1024;
static constexpr uint32_t SORT_BIG_CAP = 1u << 18;     // long buckets sorted one by one; more of them (or one beyond SORT_BIG_LONGEST): the library over all bits
static constexpr uint32_t SORT_BIG_LONGEST = 1u << 20;
static constexpr uint32_t RANK_TILE = 2048, RANK_HALO = 128;
template <class Key>
__global__ __launch_bounds__(256) void bucket_rank_kernel(const Key *__restrict__ kin, const uint32_t *__restrict__ vin, Key *__restrict__ kout,
                                                          uint32_t *__restrict__ vout, uint32_t n, int shift, uint32_t *__restrict__ big,
                                                          uint32_t *__restrict__ bigHead)
{
    // A tile of keys plus a short halo in LDS, and one bit per staged query "first of its bucket" (a ballot per 64 queries).
    // A query finds its bucket's ends in that bit mask (count-leading/trailing-zeros; longer buckets walk the mask 64 queries
    // a step), then counts: one LDS read and one compare per member.  Members of a bucket that reaches beyond the staged
    // window are read from global memory.
    constexpr uint32_t WIN = RANK_TILE + 2 * RANK_HALO;
    static_assert(WIN % 256 == 0, "whole wavefronts stage the window");
    __shared__ Key sK[WIN];
    __shared__ uint64_t sHead[WIN / 64];
    const uint32_t base = blockIdx.x * RANK_TILE;
    const uint32_t lo = base >= RANK_HALO ? base - RANK_HALO : 0u;
    const uint32_t hi = (uint64_t)base + RANK_TILE + RANK_HALO < (uint64_t)n ? base + RANK_TILE + RANK_HALO : n;
    const uint32_t win = hi - lo;
    for (uint32_t x = threadIdx.x; x < WIN; x += 256u) {
        bool head = false;
        if (x < win) {
            const uint32_t q = lo + x;
            const Key k = kin[q];
            head = q == 0u || (uint64_t)(kin[q - 1u] >> shift) != (uint64_t)(k >> shift);
            sK[x] = k;
        }
        const uint64_t m = __ballot(head);
        if ((threadIdx.x & 63u) == 0u) sHead[x >> 6] = m;
    }
    __syncthreads();
    const uint32_t lastWord = (win - 1u) >> 6;
    for (uint32_t e = threadIdx.x; e < RANK_TILE; e += 256u) {
        const uint32_t p = base + e;
        if (p >= n) break;
        const uint32_t i0 = p - lo, bit = i0 & 63u;
        const Key k = sK[i0];
        bool edge = false;
        uint32_t m = i0 >> 6;
        uint64_t w = sHead[m] & (~0ull >> (63u - bit));                   // the bucket's first member: the last flag at or before me
        while (w == 0ull && m > 0u) w = sHead[--m];
        uint32_t hs = 0;
        if (w == 0ull) edge = true; else hs = m * 64u + 63u - (uint32_t)__clzll((long long)w);
        m = i0 >> 6;
        w = sHead[m] & ((~0ull << bit) << 1);                             // its end: the first flag after me
        while (w == 0ull && m < lastWord) w = sHead[++m];
        uint32_t he = win;
        if (w == 0ull) edge = edge || (hi < n); else he = m * 64u + (uint32_t)__ffsll((unsigned long long)w) - 1u;
        uint32_t L = i0 - hs, members = he - hs, rank = 0;
        if (!edge && members <= SORT_BUCKET_LIMIT) {                      // before me: not larger; after me: smaller (= the stable order)
            for (uint32_t j = hs; j < i0; ++j) rank += (sK[j] <= k) ? 1u : 0u;
            for (uint32_t j = i0 + 1u; j < he; ++j) rank += (sK[j] < k) ? 1u : 0u;
        }
        if (edge) {                                                       // the bucket leaves the window: the whole scan again, from global memory
            const uint64_t top = (uint64_t)(k >> shift);
            uint32_t R = 0;
            rank = 0; L = 0;
            for (uint32_t q = p; q > 0 && L < SORT_BUCKET_LIMIT;) { --q; const Key o = kin[q]; if ((uint64_t)(o >> shift) != top) break; ++L; rank += (o <= k) ? 1u : 0u; }
            for (uint32_t q = p + 1; q < n && R < SORT_BUCKET_LIMIT; ++q) { const Key o = kin[q]; if ((uint64_t)(o >> shift) != top) break; ++R; rank += (o < k) ? 1u : 0u; }
            members = L + R + 1u;
        }
        // a bucket of more than SORT_BUCKET_LIMIT members stays as it is (every member decides the same way: it sees both
        // ends of the bucket or knows it is longer); its first member files it for the segmented sort that follows
        if (members > SORT_BUCKET_LIMIT) {
            kout[p] = k; vout[p] = vin[p];
            if (L == 0u) { const uint32_t at = atomicAdd(big, 1u); if (at < SORT_BIG_CAP) bigHead[at] = p; }
        } else { const uint32_t at = p - L + rank; kout[at] = k; vout[at] = vin[p]; }
    }
}

// The same for 64-bit keys, where the bits below the bucket's fit 20: LDS holds one word per query -- those bits above its
// position in the window, so members compare like (key, position) with ONE 32-bit compare -- and one bit per query "first of
// its bucket" (a ballot per 64 staged queries).  A query finds its bucket's ends in that bit mask (count-leading/trailing-zeros;
// longer buckets walk the mask 64 queries a step) and then counts the members that order before it: one LDS read, one compare
// and one add per member, no open/closed bookkeeping.  A bucket that reaches beyond the staged window is scanned in global
// memory (rare).
__global__ __launch_bounds__(256) void bucket_rank32_kernel(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout,
                                                            uint32_t *__restrict__ vout, uint32_t n, int shift, uint32_t *__restrict__ big,
                                                            uint32_t *__restrict__ bigHead)
{
    constexpr uint32_t WIN = RANK_TILE + 2 * RANK_HALO, POS_BITS = 12;
    static_assert(WIN % 256 == 0 && WIN <= (1u << POS_BITS), "window positions fit the low bits of the LDS word; whole wavefronts stage it");
    __shared__ uint32_t sW[WIN + 1];
    __shared__ uint64_t sHead[WIN / 64];
    const uint32_t base = blockIdx.x * RANK_TILE;
    const uint32_t lo = base >= RANK_HALO ? base - RANK_HALO : 0u;
    const uint32_t hi = (uint64_t)base + RANK_TILE + RANK_HALO < (uint64_t)n ? base + RANK_TILE + RANK_HALO : n;
    const uint32_t win = hi - lo;
    const uint32_t lowMask = (1u << shift) - 1u;                         // (shift + POS_BITS <= 32: checked by the caller)
    for (uint32_t x = threadIdx.x; x < WIN; x += 256u) {
        bool head = false;
        if (x < win) {
            const uint32_t q = lo + x;
            const uint64_t k = kin[q];
            head = q == 0u || (kin[q - 1u] >> shift) != (k >> shift);
            sW[x] = (((uint32_t)k & lowMask) << POS_BITS) | x;
        }
        const uint64_t m = __ballot(head);
        if ((threadIdx.x & 63u) == 0u) sHead[x >> 6] = m;
    }
    __syncthreads();
    const uint32_t lastWord = (win - 1u) >> 6;
    for (uint32_t e = threadIdx.x; e < RANK_TILE; e += 256u) {
        const uint32_t p = base + e;
        if (p >= n) break;
        const uint32_t i0 = p - lo, my = sW[i0], bit = i0 & 63u;
        bool edge = false;
        uint32_t m = i0 >> 6;
        uint64_t w = sHead[m] & (~0ull >> (63u - bit));                   // the bucket's first member: the last flag at or before me
        while (w == 0ull && m > 0u) w = sHead[--m];
        uint32_t hs = 0;
        if (w == 0ull) edge = true; else hs = m * 64u + 63u - (uint32_t)__clzll((long long)w);
        m = i0 >> 6;
        w = sHead[m] & ((~0ull << bit) << 1);                             // its end: the first flag after me
        while (w == 0ull && m < lastWord) w = sHead[++m];
        uint32_t he = win;
        if (w == 0ull) edge = edge || (hi < n); else he = m * 64u + (uint32_t)__ffsll((unsigned long long)w) - 1u;
        uint32_t L = i0 - hs, members = he - hs, rank = 0;
        if (!edge && members <= SORT_BUCKET_LIMIT) {
            for (uint32_t j = hs; j < he; j += 2u) {
                const uint32_t a = sW[j], b = sW[j + 1u];
                rank += (a < my ? 1u : 0u) + ((j + 1u < he && b < my) ? 1u : 0u);
            }
        }
        const uint64_t k = kin[p];
        if (edge) {                                                       // the bucket leaves the window: the whole scan again, from global memory
            const uint64_t top = k >> shift;
            uint32_t R = 0;
            rank = 0; L = 0;
            for (uint32_t q = p; q > 0 && L < SORT_BUCKET_LIMIT;) { --q; const uint64_t o = kin[q]; if ((o >> shift) != top) break; ++L; rank += (o <= k) ? 1u : 0u; }
            for (uint32_t q = p + 1; q < n && R < SORT_BUCKET_LIMIT; ++q) { const uint64_t o = kin[q]; if ((o >> shift) != top) break; ++R; rank += (o < k) ? 1u : 0u; }
            members = L + R + 1u;
        }
        // a bucket of more than SORT_BUCKET_LIMIT members stays as it is (every member decides the same way); its first member
        // files it for the segmented sort that follows
        if (members > SORT_BUCKET_LIMIT) {
            kout[p] = k; vout[p] = vin[p];
            if (L == 0u) { const uint32_t at = atomicAdd(big, 1u); if (at < SORT_BIG_CAP) bigHead[at] = p; }
        } else { const uint32_t at = p - L + rank; kout[at] = k; vout[at] = vin[p]; }
    }
}

// 64-bit keys with up to 40 bits below the bucket's (SORT_TOP = 32: 28): the same with one 64-bit word per query -- those bits
// above its position in the window -- so that members still compare like (key, position) with ONE compare, two members
// per step.
__global__ __launch_bounds__(256) void bucket_rank64_kernel(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout,
                                                            uint32_t *__restrict__ vout, uint32_t n, int shift, uint32_t *__restrict__ big,
                                                            uint32_t *__restrict__ bigHead)
{
    constexpr uint32_t WIN = RANK_TILE + 2 * RANK_HALO, POS_BITS = 12;
    static_assert(WIN % 256 == 0 && WIN <= (1u << POS_BITS), "window positions fit the low bits of the LDS word; whole wavefronts stage it");
    __shared__ __attribute__((aligned(16))) uint64_t sW[WIN + 2];
    __shared__ uint64_t sHead[WIN / 64];
    const uint32_t base = blockIdx.x * RANK_TILE;
    const uint32_t lo = base >= RANK_HALO ? base - RANK_HALO : 0u;
    const uint32_t hi = (uint64_t)base + RANK_TILE + RANK_HALO < (uint64_t)n ? base + RANK_TILE + RANK_HALO : n;
    const uint32_t win = hi - lo;
    const uint64_t lowMask = (1ull << shift) - 1ull;                      // (shift + POS_BITS <= 52: checked by the caller)
    for (uint32_t x = threadIdx.x; x < WIN; x += 256u) {
        bool head = false;
        if (x < win) {
            const uint32_t q = lo + x;
            const uint64_t k = kin[q];
            head = q == 0u || (kin[q - 1u] >> shift) != (k >> shift);
            sW[x] = ((k & lowMask) << POS_BITS) | x;
        }
        const uint64_t m = __ballot(head);
        if ((threadIdx.x & 63u) == 0u) sHead[x >> 6] = m;
    }
    __syncthreads();
    const uint32_t lastWord = (win - 1u) >> 6;
    for (uint32_t e = threadIdx.x; e < RANK_TILE; e += 256u) {
        const uint32_t p = base + e;
        if (p >= n) break;
        const uint32_t i0 = p - lo, bit = i0 & 63u;
        const uint64_t my = sW[i0];
        bool edge = false;
        uint32_t m = i0 >> 6;
        uint64_t w = sHead[m] & (~0ull >> (63u - bit));                   // the bucket's first member: the last flag at or before me
        while (w == 0ull && m > 0u) w = sHead[--m];
        uint32_t hs = 0;
        if (w == 0ull) edge = true; else hs = m * 64u + 63u - (uint32_t)__clzll((long long)w);
        m = i0 >> 6;
        w = sHead[m] & ((~0ull << bit) << 1);                             // its end: the first flag after me
        while (w == 0ull && m < lastWord) w = sHead[++m];
        uint32_t he = win;
        if (w == 0ull) edge = edge || (hi < n); else he = m * 64u + (uint32_t)__ffsll((unsigned long long)w) - 1u;
        uint32_t L = i0 - hs, members = he - hs, rank = 0;
        if (!edge && members <= SORT_BUCKET_LIMIT) {
            uint32_t j = hs;
            if (j & 1u) { rank += sW[j] < my ? 1u : 0u; ++j; }            // (pairs of words are read from even positions: 16-byte LDS reads)
            for (; j + 1u < he; j += 2u) {
                const ulonglong2 ab = *reinterpret_cast<const ulonglong2 *>(&sW[j]);
                rank += (ab.x < my ? 1u : 0u) + (ab.y < my ? 1u : 0u);
            }
            if (j < he) rank += sW[j] < my ? 1u : 0u;
        }
        const uint64_t k = kin[p];
        if (edge) {                                                       // the bucket leaves the window: the whole scan again, from global memory
            const uint64_t top = k >> shift;
            uint32_t R = 0;
            rank = 0; L = 0;
            for (uint32_t q = p; q > 0 && L < SORT_BUCKET_LIMIT;) { --q; const uint64_t o = kin[q]; if ((o >> shift) != top) break; ++L; rank += (o <= k) ? 1u : 0u; }
            for (uint32_t q = p + 1; q < n && R < SORT_BUCKET_LIMIT; ++q) { const uint64_t o = kin[q]; if ((o >> shift) != top) break; ++R; rank += (o < k) ? 1u : 0u; }
            members = L + R + 1u;
        }
        if (members > SORT_BUCKET_LIMIT) {
            kout[p] = k; vout[p] = vin[p];
            if (L == 0u) { const uint32_t at = atomicAdd(big, 1u); if (at < SORT_BIG_CAP) bigHead[at] = p; }
        } else { const uint32_t at = p - L + rank; kout[at] = k; vout[at] = vin[p]; }
    }
}

// [begin, end) of the listed buckets: the end by bisection over the top bits (the pairs are ordered by them)
template <class Key>
__global__ void bucket_bounds_kernel(const Key *__restrict__ kin, uint32_t n, int shift, const uint32_t *__restrict__ head, uint32_t nHead,
                                     uint32_t *__restrict__ begin, uint32_t *__restrict__ end, uint32_t *__restrict__ longest)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nHead) return;
    const uint32_t p = head[x];
    const uint64_t top = (uint64_t)(kin[p] >> shift);
    uint32_t lo = p, hi = n;                                              // first position whose top bits are larger
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)(kin[mid] >> shift) <= top) lo = mid + 1; else hi = mid; }
    begin[x] = p; end[x] = lo;
    atomicMax(longest, lo - p);
}

template <class Key>
static int sort_and_range_impl(kasa_ctx *c, int unique)
{
    c->grouped = false; c->slotOf = nullptr;
    HIPCHK(hipSetDevice(c->ix->device));
    uint64_t nQ = c->nQ;
    int rc;
    if ((rc = c->qKmerB.reserve(nQ * sizeof(Key) + 64)) || (rc = c->qReadB.reserve(nQ * 4 + 64))) return rc;
    hipEvent_t a, b;
    if ((rc = timer_begin(c, c->timers[KASA_STAGE_SORT], &a, &b))) return rc;
    if (nQ > 0) {
        const unsigned BITS = (unsigned)KeyTraits<Key>::BITS;
        auto radix = [&](DevBuf &kIn, DevBuf &vIn, DevBuf &kOut, DevBuf &vOut, unsigned lo, unsigned hi) -> int {
            size_t tmpBytes = 0;
            HIPCHK(rocprim::radix_sort_pairs(nullptr, tmpBytes, kIn.as<Key>(), kOut.as<Key>(), vIn.as<uint32_t>(), vOut.as<uint32_t>(), (size_t)nQ, lo, hi, c->stream));
            int rc2 = c->sortTmp.reserve(tmpBytes);
            if (rc2) return rc2;
            HIPCHK(rocprim::radix_sort_pairs(c->sortTmp.p, tmpBytes, kIn.as<Key>(), kOut.as<Key>(), vIn.as<uint32_t>(), vOut.as<uint32_t>(), (size_t)nQ, lo, hi, c->stream));
            return KASA_OK;
        };
        if (c->debugFlags & 64) {                                       // test tap: the library sort over all key bits
            if ((rc = radix(c->qKmerA, c->qReadA, c->qKmerB, c->qReadB, 0u, BITS))) return rc;
        } else {
            // radix passes over the top 32 (40) bits only (4 of 8 resp. 5 of 16 passes), then every query finds its place
            // inside its bucket (bucket_rank_kernel)
            // 64-bit keys: four passes and buckets of 28 bits' worth; 128-bit keys: five passes and buckets of 85 bits' worth, the
            // form of rounds 2-3 (measured at C3: a pass costs 12.6 ms, the longer buckets with 128-bit compares 14.4).  Test
            // tap 2097152: the other choice.
            const unsigned top = ((sizeof(Key) == 8) != ((c->debugFlags & 2097152) != 0)) ? SORT_TOP : SORT_TOP_OLD;
            uint32_t *big = c->misc.as<uint32_t>() + 43, *longest = c->misc.as<uint32_t>() + 44;
            if ((rc = c->sortBig.reserve((size_t)SORT_BIG_CAP * 12 + 64))) return rc;
            uint32_t *bigHead = c->sortBig.as<uint32_t>(), *segBegin = bigHead + SORT_BIG_CAP, *segEnd = segBegin + SORT_BIG_CAP;
            if (c->debugFlags & 512) {                                  // test tap: the library's passes over the same bits
                if ((rc = radix(c->qKmerA, c->qReadA, c->qKmerB, c->qReadB, BITS - top, BITS))) return rc;
            } else {
                // the hand-written passes (kasa_radix.h): A -> B -> A ... ; four passes end in A, which then takes B's name
                if ((rc = c->sortTmp.reserve(kasa_radix::scratch_bytes<Key>(nQ)))) return rc;
                Key *kRes; uint32_t *vRes;
                hipEvent_t ka, kb;
                if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_SORT_PASSES], &ka, &kb))) return rc;
                HIPCHK(kasa_radix::sort_pairs<Key>(c->qKmerA.as<Key>(), c->qReadA.as<uint32_t>(), c->qKmerB.as<Key>(), c->qReadB.as<uint32_t>(), (uint32_t)nQ,
                                                   (int)(BITS - top), (int)top, c->sortTmp.p, c->stream, &kRes, &vRes, (c->debugFlags & 524288) ? kasa_radix::MODE_FIRST : 0));   // (test tap 524288: the look-back before the keys are ordered in LDS)
                if ((rc = timer_end(c, c->kernels[KASA_KERNEL_SORT_PASSES], ka, kb))) return rc;
                if (kRes == c->qKmerA.as<Key>() && (top / 8) % 2 == 0) { std::swap(c->qKmerA, c->qKmerB); std::swap(c->qReadA, c->qReadB); }   // (an even number of passes ends where it began)
                if (kRes != c->qKmerB.as<Key>()) return fail(KASA_E_HIP, "query sort: unexpected result buffer");
            }
            HIPCHK(hipMemsetAsync(big, 0, 8, c->stream));
            hipEvent_t ra, rb;
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_BUCKET_RANK], &ra, &rb))) return rc;
            static_assert(KeyTraits<uint64_t>::BITS - SORT_TOP_OLD + 12 <= 32, "bucket_rank32_kernel: low key bits and window position share a word");
            static_assert(KeyTraits<uint64_t>::BITS - SORT_TOP + 12 <= 52, "bucket_rank64_kernel: low key bits and window position share a word");
            if (sizeof(Key) == 8 && top != SORT_TOP_OLD && !(c->debugFlags & 4194304))   // (test tap 4194304: the kernel for any key width)
                bucket_rank64_kernel<<<blocks_for(nQ, RANK_TILE), 256, 0, c->stream>>>(c->qKmerB.as<uint64_t>(), c->qReadB.as<uint32_t>(), c->qKmerA.as<uint64_t>(),
                                                                                      c->qReadA.as<uint32_t>(), (uint32_t)nQ, (int)(BITS - top), big, bigHead);
            else if (sizeof(Key) == 8 && top != SORT_TOP_OLD)
                bucket_rank_kernel<Key><<<blocks_for(nQ, RANK_TILE), 256, 0, c->stream>>>(c->qKmerB.as<Key>(), c->qReadB.as<uint32_t>(), c->qKmerA.as<Key>(),
                                                                                          c->qReadA.as<uint32_t>(), (uint32_t)nQ, (int)(BITS - top), big, bigHead);
            else if constexpr (sizeof(Key) == 8)
                bucket_rank32_kernel<<<blocks_for(nQ, RANK_TILE), 256, 0, c->stream>>>(c->qKmerB.as<uint64_t>(), c->qReadB.as<uint32_t>(), c->qKmerA.as<uint64_t>(),
                                                                                      c->qReadA.as<uint32_t>(), (uint32_t)nQ, (int)(BITS - top), big, bigHead);
            else
                bucket_rank_kernel<Key><<<blocks_for(nQ, RANK_TILE), 256, 0, c->stream>>>(c->qKmerB.as<Key>(), c->qReadB.as<uint32_t>(), c->qKmerA.as<Key>(),
                                                                                          c->qReadA.as<uint32_t>(), (uint32_t)nQ, (int)(BITS - top), big, bigHead);
            HIPCHK(hipGetLastError());
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_BUCKET_RANK], ra, rb))) return rc;
            uint32_t hBig = 0;
            HIPCHK(hipMemcpyAsync(&hBig, big, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (hBig) {
                // buckets too long to rank by counting (an input that repeats itself) were copied through as they are:
                // each of them is sorted by the remaining bits on its own (segmented radix sort, B -> A in place of the
                // copy).  Too many of them, or one too long for a single workgroup: the library over all bits instead --
                // B is a stable rearrangement of the input, so its full sort is the full sort of the input.
                uint32_t hLongest = 0;
                if (hBig <= SORT_BIG_CAP) {
                    bucket_bounds_kernel<Key><<<blocks_for(hBig, 256), 256, 0, c->stream>>>(c->qKmerB.as<Key>(), (uint32_t)nQ, (int)(BITS - top), bigHead, hBig,
                                                                                         segBegin, segEnd, longest);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemcpyAsync(&hLongest, longest, 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(hipStreamSynchronize(c->stream));
                }
                if (hBig <= SORT_BIG_CAP && hLongest <= SORT_BIG_LONGEST && !(c->debugFlags & 128)) {   // (test tap 128: the last resort)
                    size_t tmpBytes = 0;
                    HIPCHK(rocprim::segmented_radix_sort_pairs(nullptr, tmpBytes, c->qKmerB.as<Key>(), c->qKmerA.as<Key>(), c->qReadB.as<uint32_t>(), c->qReadA.as<uint32_t>(),
                                                               (unsigned)nQ, hBig, segBegin, segEnd, 0u, BITS - top, c->stream));
                    if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
                    HIPCHK(rocprim::segmented_radix_sort_pairs(c->sortTmp.p, tmpBytes, c->qKmerB.as<Key>(), c->qKmerA.as<Key>(), c->qReadB.as<uint32_t>(), c->qReadA.as<uint32_t>(),
                                                               (unsigned)nQ, hBig, segBegin, segEnd, 0u, BITS - top, c->stream));
                } else if ((rc = radix(c->qKmerB, c->qReadB, c->qKmerA, c->qReadA, 0u, BITS))) return rc;
            }
            std::swap(c->qKmerA, c->qKmerB);                            // the sorted pairs are in "B" again
            std::swap(c->qReadA, c->qReadB);
        }
    }
    c->qKmer = c->qKmerB.p;
    c->qRead = c->qReadB.as<uint32_t>();
    if (c->payloadIsSlot && nQ > 0) {
        if (unique) {                                                  // -e compares read ids: back from slots to reads
            slot_to_read_kernel<<<blocks_for(nQ, 256), 256, 0, c->stream>>>(c->qReadB.as<uint32_t>(), (uint32_t)nQ, c->kmerOff.as<uint64_t>(), (uint32_t)c->nReads, c->qReadA.as<uint32_t>());
            HIPCHK(hipMemcpyAsync(c->qReadB.p, c->qReadA.p, nQ * 4, hipMemcpyDeviceToDevice, c->stream));
            c->payloadIsSlot = false;
        } else c->slotOf = c->qReadB.as<uint32_t>();                   // the payload of the sort IS the slot
    }
    if (unique && nQ > 1) {
        // -e (Compare.hpp:3167-3178): drop records equal in (k-mer, read id) to their predecessor.  The sort above is
        // stable and the encoder emits reads in ascending order, so every duplicate of a read is adjacent here.
        if ((rc = c->rep.reserve(nQ * 4 + 64))) return rc;            // rep is free until the lookup: slots of the survivors
        uint32_t *slot = c->rep.as<uint32_t>();
        unique_flag_kernel<Key><<<blocks_for(nQ, 256), 256, 0, c->stream>>>(c->keys<Key>(), c->qRead, (uint32_t)nQ, slot);
        size_t tmpBytes = 0;
        HIPCHK(rocprim::inclusive_scan(nullptr, tmpBytes, slot, slot, (size_t)nQ, rocprim::plus<uint32_t>(), c->stream));
        if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
        HIPCHK(rocprim::inclusive_scan(c->sortTmp.p, tmpBytes, slot, slot, (size_t)nQ, rocprim::plus<uint32_t>(), c->stream));
        unique_scatter_kernel<Key><<<blocks_for(nQ, 256), 256, 0, c->stream>>>(c->keys<Key>(), c->qRead, (uint32_t)nQ, slot,
            c->qKmerA.as<Key>(), c->qReadA.as<uint32_t>());
        HIPCHK(hipGetLastError());
        uint32_t kept = 0;
        HIPCHK(hipMemcpyAsync(&kept, slot + (nQ - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        nQ = c->nQ = kept; c->uniqueDone = true;
        // back into the "sorted" buffers: later stages reuse the A buffers as scratch
        HIPCHK(hipMemcpyAsync(c->qKmerB.p, c->qKmerA.p, nQ * sizeof(Key), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->qReadB.p, c->qReadA.p, nQ * 4, hipMemcpyDeviceToDevice, c->stream));
        // k-mers per read changed: recount, kmerOff = exclusive running sum (what the score kernels index plist with)
        uint64_t *ko = c->kmerOff.as<uint64_t>();
        HIPCHK(hipMemsetAsync(ko, 0, ((size_t)c->nReads + 1) * 8, c->stream));
        read_count_kernel<<<blocks_for(nQ, 256), 256, 0, c->stream>>>(c->qReadB.as<uint32_t>(), (uint32_t)nQ, ko);
        HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, ko, ko, (uint64_t)0, (size_t)c->nReads + 1, rocprim::plus<uint64_t>(), c->stream));
        if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
        HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, ko, ko, (uint64_t)0, (size_t)c->nReads + 1, rocprim::plus<uint64_t>(), c->stream));
    }
    if ((rc = timer_end(c, c->timers[KASA_STAGE_SORT], a, b))) return rc;
    return lookup_part<Key>(c);
}

// depth and representative of every sorted query, first closing positions per tile: the batch is "sorted and ranged"
template <class Key>
static int lookup_part(kasa_ctx *c)
{
    const uint64_t nQ = c->nQ;
    int rc;
    hipEvent_t a, b;
    if ((rc = c->depth.reserve(nQ + 64)) || (rc = c->rep.reserve(nQ * 4 + 64))) return rc;
    const uint32_t nTiles = (uint32_t)((nQ + TILE - 1) / TILE);
    if ((rc = c->tileFirst.reserve((size_t)c->nK * (nTiles + 1) * 4)) || (rc = c->tileNext.reserve((size_t)c->nK * (nTiles + 1) * 4)) ||
        (rc = c->tileBounds.reserve(((size_t)nTiles + 1) * 8)))
        return rc;

    if ((rc = timer_begin(c, c->timers[KASA_STAGE_LOOKUP], &a, &b))) return rc;
    if (nQ > 0) {
        hipEvent_t ka, kb;
        if (c->lookupMode == 1) {
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_LOOKUP], &ka, &kb))) return rc;
            lookup_kernel<Key><<<nTiles, TILE_THREADS, 0, c->stream>>>(c->keys<Key>(), (uint32_t)nQ, c->ix->kmer.as<Key>(), (uint32_t)c->ix->n,
                c->ix->table.as<uint32_t>(), c->ix->tb, c->kHigh, c->kLow, c->depth.as<uint8_t>(), c->rep.as<uint32_t>(),
                c->tileFirst.as<uint32_t>(), nTiles);
        } else {
            tile_bounds_kernel<Key><<<blocks_for(2ull * nTiles, 256), 256, 0, c->stream>>>(c->keys<Key>(), (uint32_t)nQ, c->ix->kmer.as<Key>(),
                c->ix->table.as<uint32_t>(), c->ix->tb, nTiles, c->tileBounds.as<uint32_t>());
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_LOOKUP], &ka, &kb))) return rc;   // lookup_tile_kernel alone
            lookup_tile_kernel<Key><<<nTiles, TILE_THREADS, 0, c->stream>>>(c->keys<Key>(), (uint32_t)nQ, c->ix->kmer.as<Key>(), (uint32_t)c->ix->n,
                c->ix->table.as<uint32_t>(), c->ix->tb, c->tileBounds.as<uint32_t>(), c->kHigh, c->kLow, c->depth.as<uint8_t>(),
                c->rep.as<uint32_t>(), c->tileFirst.as<uint32_t>(), nTiles);
        }
        HIPCHK(hipGetLastError());
        if ((rc = timer_end(c, c->kernels[KASA_KERNEL_LOOKUP], ka, kb))) return rc;
        if ((rc = tile_suffix(c, nTiles))) return rc;
    }
    if ((rc = timer_end(c, c->timers[KASA_STAGE_LOOKUP], a, b))) return rc;
    c->state = 3;
    return KASA_OK;
}

extern "C" int kasa_batch_sort_and_range(kasa_ctx *c, int unique)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (c->state < 2) return fail(KASA_E_STATE, "kasa_batch_sort_and_range: batch not encoded");
    return c->ix->wide ? sort_and_range_impl<key128>(c, unique) : sort_and_range_impl<uint64_t>(c, unique);
}

static constexpr int GITEMS = 2;                  // the group kernel gives a thread two queries of the tile
static constexpr int GTHREADS = TILE / GITEMS;    // 512
static constexpr int GOVF = 1024;                 // segments beyond the inline ones a tile can park in LDS (narrow records)
static constexpr int GSPAN = 3072;                // index entries around the tile's matches staged in LDS (taxon + neighbour counts)
static constexpr uint32_t GMARGIN = 96;           // ... this many beyond the first and last representative

// ------------------------------------------------------------------------------------------------
// event records
// ------------------------------------------------------------------------------------------------
// One record per query, RW 32-bit words (RW = 8: up to 8 levels, RW = 16: up to 25), written to the query's SLOT: reads
// in batch order, the queries of a read in sorted order -- so the score kernels stream a read's records front to back.
//   [0] p      sorted position of the query
//   [1] Fmax   the last flush position of its groups (all of them are closed once the stream reaches it)
//   [2] d | order << 5    d = deepest matched k (0: no match, nothing else is valid); order (RW = 8) = the levels
//                         lv = kHigh - k of its events in flush order (F ascending, k ascending), 3 bits each;
//                         bits 29, 30 (RW = 8): REC_SPLIT, REC_SAT
//   [3] nseg   number of taxon segments (RW = 8: low byte, 255 = "255 or more"; the upper 24 bits hold |T_k| of the
//                 levels lv = 0..7, 3 bits each, 7 = "7 or more")
//   RW = 8 : [4..7]  the segments when there are at most 4; else 3 segments and [7] = pool offset of the others
//   RW = 16: [4..7]  order, 5 bits per event (a 128-bit value, first event in the low bits);  [8..15] the segments when
//            there are at most 8; else 7 segments and [15] = pool offset of the others
//   pool block: {nseg, [sizes: 4 words, only with REC_SAT], segments INL-1 ...}.  Segments come in descending order of
//   their last level: the taxa the query really matches first, chance matches of its short prefixes last.  sizes: the
//   exact |T_k| of the levels lv = 0..7, 16 bits each (two per word) -- the 3-bit counts of word [3] stop at 7.
// A segment is one index entry of the query's kLow-group seen from the query: taxon | kFirst << 22 | kLast << 27 -- the
// entry puts its taxon into the taxon set T_k of the levels kFirst..kLast (kLast = letters it shares with the query,
// capped at d; kFirst - 1 = letters it shares with the nearest earlier entry of the same taxon, which represents the
// taxon up to there).  |T_k| = number of segments covering k.  This replaces the reference's per-level sBitArray sets
// (Compare.hpp:917-955, BitArray.hpp:98-117) for all levels of a query at once.
static constexpr uint32_t SEG_TAX_MASK = (1u << 22) - 1u;
static constexpr uint32_t REC_SPLIT = 1u << 29;   // word [2]: some segment starts above kLow, i.e. a taxon may own several segments
static constexpr uint32_t REC_SAT = 1u << 30;     // RW = 8, word [2]: some level has 7 or more taxa (its 3-bit count is saturated)
template <int RW> struct RecTraits;
template <> struct RecTraits<8> { static constexpr int LEVELS = 8, INL = 4, SEG0 = 4, OBITS = 3; };
template <> struct RecTraits<16> { static constexpr int LEVELS = 25, INL = 8, SEG0 = 8, OBITS = 5; };
__device__ __forceinline__ bool seg_covers(uint32_t s, uint32_t k) { return ((s >> 22) & 31u) <= k && k <= (s >> 27); }
// levels of a segment as a mask over lv = kHigh - k
__device__ __forceinline__ uint32_t seg_level_mask(uint32_t s, int kHigh)
{
    const int lvLo = kHigh - (int)(s >> 27), lvHi = kHigh - (int)((s >> 22) & 31u);      // kLast -> lowest lv
    return ((2u << lvHi) - 1u) & ~((1u << lvLo) - 1u);
}
// record readers (w = the record's words)
template <int RW> __device__ __forceinline__ uint32_t rec_nseg(const uint32_t *w, const uint32_t *__restrict__ pool)
{
    if constexpr (RW == 8) { const uint32_t n = w[3] & 255u; return n == 255u ? pool[w[7]] : n; }
    else return w[3];
}
static constexpr uint32_t POOL_SIZES = 4;         // words of exact level sizes in the pool block of a REC_SAT record
template <int RW> __device__ __forceinline__ uint32_t rec_seg(const uint32_t *w, const uint32_t *__restrict__ pool, uint32_t nseg, uint32_t i)
{
    typedef RecTraits<RW> RT;
    if (nseg <= (uint32_t)RT::INL || i < (uint32_t)RT::INL - 1u) return w[RT::SEG0 + i];
    const uint32_t skip = (RW == 8 && (w[2] & REC_SAT)) ? POOL_SIZES : 0u;
    return pool[w[RT::SEG0 + RT::INL - 1] + 1u + skip + i - ((uint32_t)RT::INL - 1u)];
}

// levels (bit lv = kHigh - k) at which sorted position p closes the groups before it: a new prefix range closes all of
// them, otherwise p opens a new matched group at the levels ql < k <= d (ql = letters shared with its predecessor)
__device__ __forceinline__ uint32_t special_mask(int ql, int d, int kHigh, uint32_t allLv)
{
    if (ql < RANGE_LETTERS) return allLv;
    return (d > ql) ? ((((2u << (d - ql - 1)) - 1u) << (kHigh - d)) & allLv) : 0u;
}

__device__ __forceinline__ uint32_t block_excl_prefix_sum(uint32_t v, uint32_t *sh, uint32_t &total)
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t incl = v;
    incl = wave_incl_sum(incl);
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < GTHREADS / 64; ++w) { const uint32_t o = sh[w]; if (w < wv) before += o; all += o; }
    total = all;
    __syncthreads();
    return before + incl - v;
}

// The taxon segments of one query: walk the index outwards from `j` (an entry sharing the query's deepest matched
// prefix) as long as the entries share the kLow-group's letters with the query.  Letters entry i shares with the query
// = min(d, letters shared by all neighbours between i and j) -- `meta` holds the neighbour counts.  Both directions are
// merged so that the segments come in descending order of their last level.  Calls emit(seg).
// `meta(i)` / `tax(i)` read the index arrays (group_kernel serves them from the tile's span staged in LDS).
// maxSteps: give up (false) after that many entries beyond j -- a long walk is the whole wavefront's (coop_walk).
template <class Meta, class GetMeta, class GetTax, class Emit>
__device__ __forceinline__ bool walk_segments(uint32_t j, int d, int kLow, GetMeta meta, GetTax tax, uint32_t nIdx, Emit emit, uint32_t maxSteps = 0xFFFFFFFFu)
{
    constexpr int LM = sizeof(Meta) == 1 ? 15 : 255, DS = sizeof(Meta) == 1 ? 4 : 8;
    const int gLow = group_letters(kLow);
    auto one = [&](uint32_t i, int kLast, uint32_t m) {
        const int dup = (int)(m >> DS);
        const int kFirst = dup < RANGE_LETTERS ? kLow : (dup + 1 > kLow ? dup + 1 : kLow);
        if (kFirst <= kLast) emit(tax(i) | ((uint32_t)kFirst << 22) | ((uint32_t)kLast << 27));
    };
    const uint32_t mj = meta(j);
    one(j, d, mj);
    // j shares at least max(d, 6) letters with the query (a match needs the 6-letter range; '^' may cut d below that)
    const int chain0 = d > RANGE_LETTERS ? d : RANGE_LETTERS;
    uint32_t li = j, ri = j + 1;                                  // next to the left: li - 1; next to the right: ri
    uint32_t mli = mj;                                            // meta[li]: letters li shares with li - 1
    uint32_t mri = ri < nIdx ? (uint32_t)meta(ri) : 0u;           // meta[ri]: letters ri shares with ri - 1
    int lc = li > 0 ? ((int)(mli & LM) < chain0 ? (int)(mli & LM) : chain0) : -1;   // letters the next left entry shares with the query
    int rc = ri < nIdx ? ((int)(mri & LM) < chain0 ? (int)(mri & LM) : chain0) : -1;
    uint32_t steps = 0;
    while (lc >= gLow || rc >= gLow) {
        if (++steps > maxSteps) return false;
        if (lc >= rc) {
            --li;
            mli = meta(li);
            one(li, lc < d ? lc : d, mli);
            const int l = (int)(mli & LM);
            lc = li > 0 ? (l < lc ? l : lc) : -1;
        } else {
            one(ri, rc < d ? rc : d, mri);
            ++ri;
            if (ri < nIdx) { mri = meta(ri); const int l = (int)(mri & LM); rc = l < rc ? l : rc; } else rc = -1;
        }
    }
    return true;
}

// A LONG walk (a conserved k-mer: hundreds of index entries around the query's place) is not one lane's business: every step
// is a dependent read, and the other 63 lanes of its wavefront wait.  coop_walk gives the entries to all 64 lanes, 64 at a
// time: first those left of j (nearest first), then those to the right.  An entry's letters in common with the query are the
// running minimum of the neighbour counts between it and j (a prefix minimum over the lanes, carried from chunk to chunk),
// capped at max(d, 6); the side ends where that falls below the kLow-group's letters.  f(ok, idx, v, side) is called by ALL
// lanes for every chunk (it may hold wavefront-wide operations); `ok` marks the lanes whose entry belongs to the walk.
#define KASA_DPP_ID(v, ctrl, rows, id) __builtin_amdgcn_update_dpp((int)(id), (int)(v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ int wave_incl_min(int v)
{
    constexpr int TOP = 0x7fffffff;
    { const int o = KASA_DPP_ID(v, 0x111, 0xf, TOP); v = o < v ? o : v; }
    { const int o = KASA_DPP_ID(v, 0x112, 0xf, TOP); v = o < v ? o : v; }
    { const int o = KASA_DPP_ID(v, 0x114, 0xf, TOP); v = o < v ? o : v; }
    { const int o = KASA_DPP_ID(v, 0x118, 0xf, TOP); v = o < v ? o : v; }
    { const int o = KASA_DPP_ID(v, 0x142, 0xa, TOP); v = o < v ? o : v; }
    { const int o = KASA_DPP_ID(v, 0x143, 0xc, TOP); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_max_int(int v) { return ~__builtin_amdgcn_readlane(wave_incl_min(~v), 63); }   // (every lane gets it)
template <class Meta, class F>
__device__ __forceinline__ void coop_walk(const Meta *__restrict__ meta, const uint32_t *__restrict__ tax, uint32_t nIdx, uint32_t j, int d, int kLow, int lane, F f)
{
    // f(ok, idx, v, side, m, tx): m = meta[idx], tx = tax[idx] -- loaded here, ONE CHUNK AHEAD: where a chunk lies does not
    // depend on what the chunk before held (only whether it is needed does), so its three loads leave before the chunk before
    // is looked at; with the loads inside f every chunk was two dependent round trips to memory (a list of a conserved k-mer
    // lies beyond the tile's LDS span), five sweeps over every long list: most of the cooperative kernel's time.
    constexpr int LM = sizeof(Meta) == 1 ? 15 : 255;
    const int gLow = group_letters(kLow);
    const int chain0 = d > RANGE_LETTERS ? d : RANGE_LETTERS;
    const uint32_t lastIdx = nIdx - 1u;
    int run = chain0;
    {   // left: entry j - a shares with the query what j - a + 1 .. j share with their predecessors
        auto load = [&](uint32_t a0, uint32_t &mc, uint32_t &mo, uint32_t &tx) {
            const uint32_t a = a0 + (uint32_t)lane;
            const uint32_t idx = a <= j ? j - a : 0u;
            mc = (uint32_t)meta[idx < lastIdx ? idx + 1u : lastIdx]; mo = (uint32_t)meta[idx]; tx = tax[idx];
        };
        uint32_t mc, mo, tx, mcN = 0, moN = 0, txN = 0;
        load(1u, mc, mo, tx);
        for (uint32_t a0 = 1;; a0 += 64) {
            load(a0 + 64u, mcN, moN, txN);
            const uint32_t a = a0 + (uint32_t)lane;
            const bool valid = a <= j;
            const uint32_t idx = valid ? j - a : 0u;
            int v = wave_incl_min(valid ? (int)(mc & (uint32_t)LM) : -1);
            v = v < run ? v : run;
            const bool ok = valid && v >= gLow;
            f(ok, idx, v, 0, mo, tx);
            if (__ballot(ok) != ~0ull) break;
            run = __builtin_amdgcn_readlane(v, 63);
            mc = mcN; mo = moN; tx = txN;
        }
    }
    run = chain0;
    {   // right: entry j + b shares what j + 1 .. j + b share with their predecessors
        auto load = [&](uint32_t b0, uint32_t &mo, uint32_t &tx) {
            const uint32_t idx0 = j + b0 + (uint32_t)lane;
            const uint32_t idx = (idx0 < nIdx && idx0 > j) ? idx0 : 0u;
            mo = (uint32_t)meta[idx]; tx = tax[idx];
        };
        uint32_t mo, tx, moN = 0, txN = 0;
        load(1u, mo, tx);
        for (uint32_t b0 = 1;; b0 += 64) {
            load(b0 + 64u, moN, txN);
            const uint32_t idx = j + b0 + (uint32_t)lane;
            const bool valid = idx < nIdx && idx > j;
            int v = wave_incl_min(valid ? (int)(mo & (uint32_t)LM) : -1);
            v = v < run ? v : run;
            const bool ok = valid && v >= gLow;
            f(ok, idx, v, 1, mo, tx);
            if (__ballot(ok) != ~0ull) break;
            run = __builtin_amdgcn_readlane(v, 63);
            mo = moN; tx = txN;
        }
    }
}
// Entries a query's walk may visit lane by lane before its wavefront takes the list (coop_walk).  Measured (KASA_LONG_STEPS): on a
// crowded index, where lists are either a handful of entries or a clade's 50-200, every lane-by-lane step of a long list is
// lost (the wavefront waits for it and the list is then walked again): 147 / 128 / 121 / 119 / 119 / 122 / 124 ms at 48 / 24 / 12 /
// 8 / 6 / 4 / 3 for the 2 M-read batch; on the tiles group2_kernel lists at C2 (heavy 7-letter groups of 20-40 entries) the
// lane walk is the cheaper one: 60.0 / 66.9 / 70.9 ms at 48 / 12 / 6.  So: 8 for a context on the cooperative kernel for
// good, 48 for listed tiles.
static constexpr uint32_t LONG_STEPS = 48, LONG_STEPS_CROWDED = 8;

// Profile key of a record: {level | |T| | taxon | hits:16}.  The three upper fields are as wide as the batch needs
// (taxon: enough bits that the all-ones value is no taxon -- it marks unused slots; |T| <= 8191 and < nTaxa), so the
// radix sort of the keys runs over as few bits as possible: 27 instead of 38 for 1400 taxa and 6 levels.
struct ProfLayout {
    uint32_t tb, nb, lb;
    __host__ __device__ uint32_t bits() const { return tb + nb + lb; }
};
static inline ProfLayout prof_layout(uint32_t nTaxa, int nK)
{
    ProfLayout L;
    L.tb = 1; while ((1u << L.tb) <= nTaxa) ++L.tb;              // 2^tb > nTaxa: the all-ones taxon is free
    L.nb = L.tb < 13 ? L.tb : 13;
    L.lb = 1; while ((1u << L.lb) < (uint32_t)nK) ++L.lb;
    return L;
}
__device__ __forceinline__ uint64_t profile_key_of(uint32_t lv, uint32_t n, uint32_t tax, uint32_t hits, ProfLayout L)
{
    const uint64_t f = ((uint64_t)lv << (L.tb + L.nb)) | ((uint64_t)n << L.tb) | tax;
    return (f << 16) | hits;
}
// c / n added to a 64.64 fixed-point cell kept as three u64 accumulators {hi, mid, lo}: the 128-bit term
// x = c * floor(2^64 / n) is split into hi = x >> 64 and the two 32-bit halves of its low word, each added
// with a fire-and-forget integer atomic (value = hi + (mid * 2^32 + lo) / 2^64; mid and lo absorb up to 2^32
// terms before they could wrap).  Exact, associative, independent of the order in which waves arrive.
__device__ __forceinline__ void fixed_add(uint64_t *hiTab, uint64_t *midTab, uint64_t *loTab, size_t cell, uint64_t c, uint32_t n)
{
    if (n == 1) { atomicAdd((unsigned long long *)&hiTab[cell], (unsigned long long)c); return; }
    uint64_t R = 0xFFFFFFFFFFFFFFFFull / n;
    if ((n & (n - 1)) == 0) R += 1;                                      // n divides 2^64
    const uint64_t lo64 = c * R;
    const uint64_t hi = __umul64hi(c, R);
    const uint64_t lo = lo64 & 0xFFFFFFFFull, mid = lo64 >> 32;
    if (lo) atomicAdd((unsigned long long *)&loTab[cell], (unsigned long long)lo);
    if (mid) atomicAdd((unsigned long long *)&midTab[cell], (unsigned long long)mid);
    if (hi) atomicAdd((unsigned long long *)&hiTab[cell], (unsigned long long)hi);
}

// |T_k| of a level and the decoded record of one query, for both record widths
// |T_k| tables of wide records in LDS: TWO levels per 32-bit word, 16 bits each (a query has fewer than 2^13 segments), so
// that the tables -- which limit the resident wavefronts of the wide score kernels -- take half the room.  The +1 / -1 marks
// at the ends of a segment's level range go into a word as +-1 or +-65536 (plain or atomic adds); a borrow out of the low
// half is undone when the running sum reads the marks back as two SIGNED halves.
__device__ __forceinline__ uint32_t lvp_rows(int levels) { return (uint32_t)(levels + 3) / 2u; }   // levels + 1 marks (one past the last level)
__device__ __forceinline__ uint32_t lvp_unit(int lv) { return (lv & 1) ? 0x10000u : 1u; }
__device__ __forceinline__ uint32_t lvp_get(const uint32_t *col, int stride, int lv) { return (col[(lv >> 1) * stride] >> (16 * (lv & 1))) & 0xFFFFu; }
// marks -> sizes, in place; calls f(lv, size) for every level
template <class F> __device__ __forceinline__ void lvp_running(uint32_t *col, int stride, int nK, F f)
{
    uint32_t running = 0;
    for (int w = 0; 2 * w < nK; ++w) {
        const uint32_t x = col[w * stride];
        const int lo = (int)(int16_t)(x & 0xFFFFu);
        const int hi = (int)(int16_t)((x - (uint32_t)lo) >> 16);
        const uint32_t a = running + (uint32_t)lo, b = a + (uint32_t)hi;
        running = b;
        col[w * stride] = (a & 0xFFFFu) | (b << 16);
        f(2 * w, a);
        if (2 * w + 1 < nK) f(2 * w + 1, b);
    }
}

// Narrow records (up to 8 levels): the profile is made by group_kernel (keys of the groups' first queries) and the per-read
// kernels carry no profile records, counters or keys.  Wide records still take theirs from the per-read side.
template <int RW> struct GpOf { static constexpr bool v = RW == 8; };
// Profile keys of group_kernel ("group keys"): ONE key for a run of levels of a taxon segment that share |T| and the hits
// -- the rule: a k-mer run matched down to k = 12 against one taxon is one key, not six.
//   hits:16 | taxon:22 | |T|:13 | first level (lv = kHigh - k):5 | levels - 1:5
__device__ __forceinline__ uint64_t group_key(uint32_t lvLo, uint32_t lvHi, uint32_t n, uint32_t tax, uint32_t hits)
{
    return (uint64_t)hits | ((uint64_t)tax << 16) | ((uint64_t)n << 38) | ((uint64_t)lvLo << 51) | ((uint64_t)(lvHi - lvLo) << 56);
}

#define LDS_WAVE_SYNC_G() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// group: one workgroup per tile of TILE sorted queries, a thread owns GITEMS consecutive ones.
//   1. flush positions F_k(p) = next position after p that closes level k: inside the wavefront from two ballots per
//      level, across wavefronts through one LDS table (one barrier), across tiles from tileNext;
//   2. the flush order of the query's events and Fmax;
//   3. the taxon segments (walk_segments), the first INL of them inline, longer lists in the pool with one allocation
//      per workgroup;
//   4. the record goes to rec[slot]: slotOf[p], or p itself when slotOf is NULL (records exported in sorted order).
// COOP (narrow records): long lists are counted and placed by whole wavefronts (coop_walk).  The lean form walks every list
// lane by lane and only REPORTS a list of 255 or more segments (*needCoop; its 8-bit level counts would wrap): the host then
// groups the batch again with COOP -- and stays with it for the context's further batches, as it does when the taxon lists of
// a batch are long on average.  Data without crowded k-mers never pays for the cooperative code's registers.
template <int RW, class Key, int NKT, bool COOP = false>           // NKT: kHigh - kLow + 1 when known at compile time, else 0
__global__ __launch_bounds__(GTHREADS, (RW == 8 && COOP) ? 6 : 1) void group_kernel(
    const Key *__restrict__ qKmer, const uint8_t *__restrict__ depth, const uint32_t *__restrict__ rep,
    const uint32_t *__restrict__ slotOf, uint32_t nQ, const uint32_t *__restrict__ tileNext, uint32_t nTiles,
    const typename KeyTraits<Key>::Meta *__restrict__ meta, const uint32_t *__restrict__ tax, uint32_t nIdx, int kHigh, int kLow,
    uint32_t *__restrict__ rec, uint32_t *__restrict__ pool, uint32_t poolCap, unsigned long long *__restrict__ poolCursor, int flags,
    uint64_t *__restrict__ cntTotal, uint32_t nTaxa,
    uint64_t *__restrict__ profKeys, uint32_t keyCap, unsigned long long *__restrict__ keyCursor, ProfLayout PL,
    uint64_t *__restrict__ cntAllHi, uint64_t *__restrict__ cntAllMid, uint64_t *__restrict__ cntAllLo, uint32_t *__restrict__ needCoop,
    const uint32_t *__restrict__ tileList, uint32_t cellW)
{
    const uint32_t tileId = tileList ? (tileList[blockIdx.x] & 0x0FFFFFFFu) : blockIdx.x;   // (the tiles group2_kernel left to this one; the top bits say why)
    const int coverage = flags & 1;                                // bit 1: every query walks the index itself (test tap); bit 2: no LDS span for 64-byte records
    typedef RecTraits<RW> RT;
    constexpr int NL = NKT ? NKT : RT::LEVELS, INL = RT::INL;       // levels the unrolled loops run over
    __shared__ uint32_t shU[GTHREADS / 64];
    __shared__ uint32_t sFirst[GTHREADS / 64][NL];                 // first closing position of a wavefront, per level
    __shared__ uint32_t sBase, sBaseK;
    // the index entries the tile's walks visit (their taxa and neighbour counts), staged once: the walks are chains of
    // dependent reads, from LDS they cost tens of cycles instead of a trip to L2/HBM each
    typedef typename KeyTraits<Key>::Meta Meta;
    // 64-byte records leave through LDS: a lane owns a record, FOUR lanes store it -- one store instruction then touches 16
    // whole 64-byte cells instead of a quarter of 64 (tools/scatter_probe.hip: 19 G records/s lane by lane, 4 instructions
    // of 64 partial cells each; 49 G records/s by quads).  The output stage lies over the index span, which is dead by then.
    constexpr int SPAN_BYTES = GSPAN * 4 + GSPAN * (int)sizeof(Meta), OUT_BYTES = RW == 16 ? GTHREADS * 4 * 16 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char sRaw[SPAN_BYTES > OUT_BYTES ? SPAN_BYTES : OUT_BYTES];
    uint32_t *sTax = reinterpret_cast<uint32_t *>(sRaw);
    Meta *sMeta = reinterpret_cast<Meta *>(sRaw + GSPAN * 4);
    uint4 *sOut = reinterpret_cast<uint4 *>(sRaw);
    __shared__ uint32_t sRepLo[GTHREADS / 64], sRepHi[GTHREADS / 64];
    // narrow records: the segments beyond the inline ones wait here until the workgroup's pool block is allocated (one walk
    // per query instead of two); {segment, owner query | index in its pool list << 10}
    __shared__ uint2 sOvf[RW == 8 ? GOVF : 1];
    __shared__ uint32_t sOvfDst[RW == 8 ? TILE : 1];                 // per query of the tile: pool word of its first overflow segment
    __shared__ uint32_t sOvfN;
    // per query of the tile, for the profile keys of parked segments: hits and |T| per level (8 bits each), the levels with hits
    // and where a run of equal (|T|, hits) begins
    __shared__ unsigned long long sHitsQ[RW == 8 ? TILE : 1], sSizeQ[RW == 8 ? TILE : 1];
    __shared__ uint16_t sRunQ[RW == 8 ? TILE : 1];
    // a wavefront's scratch for its long lists: level marks / sizes [32], segments per class [64], class offsets [64]; the
    // first segments of the list (the record's inline ones)
    __shared__ uint32_t sCo[(RW == 8 && COOP) ? GTHREADS / 64 : 1][(RW == 8 && COOP) ? 160 : 1], sInl[(RW == 8 && COOP) ? GTHREADS / 64 : 1][INL];
    if (threadIdx.x == 0) sOvfN = 0u;
    const int nK = NKT ? NKT : kHigh - kLow + 1;
    const uint32_t allLv = (nK >= 32) ? 0xFFFFFFFFu : ((1u << nK) - 1u);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t base = tileId * TILE + t * GITEMS;
    int d[GITEMS];
    uint32_t rp[GITEMS], sp[GITEMS];
#pragma unroll
    for (int i = 0; i < GITEMS; ++i) {
        const uint32_t p = base + i;
        d[i] = 0; rp[i] = 0; sp[i] = 0;
        if (p < nQ) {
            const Key q = qKmer[p];
            const int ql = (p == 0) ? 0 : lcp_letters<Key>(qKmer[p - 1], q);
            d[i] = depth[p];
            rp[i] = rep[p];
            sp[i] = special_mask(ql, d[i], kHigh, allLv);
        }
    }
    // A query whose (representative, depth) are its predecessor's sees the same index entries the same way: the same
    // segments, sizes and flags.  Such FOLLOWERS (the copies of a k-mer that the reads' coverage brings, and every read of a
    // conserved region) do not walk the index and own no pool block: they take their leader's words once those are final.
    // The first query of a wavefront is a leader by construction.
    bool fol[GITEMS];
    {
        const uint32_t prevRp = (uint32_t)__shfl_up((int)rp[1], 1);
        const int prevD = __shfl_up(d[1], 1);
        const bool on = !(flags & 2);
        fol[0] = on && lane > 0 && d[0] != 0 && d[0] == prevD && rp[0] == prevRp;
        fol[1] = on && d[1] != 0 && d[1] == d[0] && rp[1] == rp[0];
    }
    const unsigned long long lead0 = __ballot(!fol[0]), lead1 = __ballot(!fol[1]);
    {   // span of the representatives (nearly monotone in p): first and last matched query of the wavefront
        const unsigned long long m0 = __ballot(d[0] != 0), m1 = __ballot(d[1] != 0);
        const unsigned long long any = m0 | m1;
        uint32_t lo = NOPOS, hi = 0;
        if (any) {
            const int lf = __ffsll((long long)any) - 1, ll = 63 - __clzll((long long)any);
            const uint32_t a0 = __shfl(rp[0], lf), a1 = __shfl(rp[1], lf), b0 = __shfl(rp[0], ll), b1 = __shfl(rp[1], ll);
            lo = ((m0 >> lf) & 1ull) ? a0 : a1;
            hi = ((m1 >> ll) & 1ull) ? b1 : b0;
        }
        if (lane == 0) { sRepLo[wv] = lo; sRepHi[wv] = hi; }
    }
    // ---- 1. flush positions
    const unsigned long long above = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    uint32_t F[GITEMS][NL];
    uint32_t open0 = 0, open1 = 0;                               // levels not closed inside the wavefront
#pragma unroll
    for (int lv = 0; lv < NL; ++lv) {
        F[0][lv] = NOPOS; F[1][lv] = NOPOS;
        if (lv >= nK) continue;
        const unsigned long long b0 = __ballot((sp[0] >> lv) & 1u), b1 = __ballot((sp[1] >> lv) & 1u);
        const unsigned long long any = b0 | b1;
        uint32_t next = NOPOS;                                    // first closing position in the lanes above
        const unsigned long long hi = any & above;
        if (hi) {
            const int l2 = __ffsll((long long)hi) - 1;
            next = tileId * TILE + (uint32_t)(wv * 64 + l2) * GITEMS + (((b0 >> l2) & 1ull) ? 0u : 1u);
        }
        F[1][lv] = next;
        F[0][lv] = ((sp[1] >> lv) & 1u) ? base + 1 : next;
        if (next == NOPOS) { open1 |= 1u << lv; if (!((sp[1] >> lv) & 1u)) open0 |= 1u << lv; }
        if (lane == 0) {
            uint32_t first = NOPOS;
            if (any) { const int l2 = __ffsll((long long)any) - 1; first = tileId * TILE + (uint32_t)(wv * 64 + l2) * GITEMS + (((b0 >> l2) & 1ull) ? 0u : 1u); }
            sFirst[wv][lv] = first;
        }
    }
    __syncthreads();
    if (open1) {
#pragma unroll
        for (int lv = 0; lv < NL; ++lv) {
            if (!((open1 >> lv) & 1u)) continue;
            uint32_t v = NOPOS;
            for (int w = wv + 1; w < GTHREADS / 64; ++w) { const uint32_t o = sFirst[w][lv]; if (o != NOPOS) { v = o; break; } }
            if (v == NOPOS) v = tileNext[(size_t)lv * nTiles + tileId];
            F[1][lv] = v;
            if ((open0 >> lv) & 1u) F[0][lv] = v;
        }
    }
    uint32_t spanLo = NOPOS, spanHi = 0;
    for (int w = 0; w < GTHREADS / 64; ++w) { spanLo = min(spanLo, sRepLo[w]); spanHi = max(spanHi, sRepHi[w]); }
    uint32_t spanN = 0;
    if (spanLo != NOPOS) {
        spanLo = spanLo > GMARGIN ? spanLo - GMARGIN : 0u;
        spanHi = min(spanHi + GMARGIN + 1u, nIdx);
        spanN = (RW == 16 && (flags & 4)) ? 0u : min(spanHi - spanLo, (uint32_t)GSPAN);   // what lies beyond is read from global memory
        for (uint32_t x = t; x < spanN; x += GTHREADS) { sTax[x] = tax[spanLo + x]; sMeta[x] = meta[spanLo + x]; }
    }
    __syncthreads();
    auto getMeta = [&](uint32_t i) -> uint32_t { const uint32_t x = i - spanLo; return x < spanN ? (uint32_t)sMeta[x] : (uint32_t)meta[i]; };
    auto getTax = [&](uint32_t i) -> uint32_t { const uint32_t x = i - spanLo; return x < spanN ? sTax[x] : tax[i]; };
    // ---- 2. + 3. per query: order of its events, taxon segments
    uint32_t w2[GITEMS], w3[GITEMS], fmax[GITEMS], nseg[GITEMS], seg[GITEMS][INL];
    bool isLong[GITEMS] = {false, false};                        // the query's walk exceeded LONG_STEPS entries
    unsigned long long cnt8x[GITEMS] = {0ull, 0ull};
    unsigned __int128 ord[GITEMS];
    uint32_t need = 0;
#pragma unroll
    for (int i = 0; i < GITEMS; ++i) {
        w2[i] = 0; w3[i] = 0; fmax[i] = 0; nseg[i] = 0;
#pragma unroll
        for (int s = 0; s < INL; ++s) seg[i][s] = 0;
        ord[i] = 0;
        if (d[i] == 0) continue;
        const int lvTop = kHigh - d[i];                           // events: levels lvTop .. nK-1
        uint32_t fm = 0;
        if constexpr (RW == 8) {
            // rank of an event = events flushed before it: smaller F, or equal F and smaller k (larger lv); one comparison per pair
            uint32_t rank[NL];
#pragma unroll
            for (int lv = 0; lv < NL; ++lv) rank[lv] = 0;
#pragma unroll
            for (int a = 0; a < NL; ++a)
#pragma unroll
                for (int b2 = a + 1; b2 < NL; ++b2) {
                    const bool both = a >= lvTop && b2 < nK;
                    const bool aFirst = F[i][a] < F[i][b2];       // a tie goes to b2 (the smaller k)
                    rank[b2] += (both && aFirst) ? 1u : 0u;
                    rank[a] += (both && !aFirst) ? 1u : 0u;
                }
#pragma unroll
            for (int lv = 0; lv < NL; ++lv) {
                if (lv < lvTop || lv >= nK) continue;
                const uint32_t f = F[i][lv];
                if (f > fm) fm = f;
                ord[i] |= (unsigned __int128)((uint32_t)lv << (RT::OBITS * rank[lv]));
            }
        } else {
#pragma unroll
            for (int lv = 0; lv < NL; ++lv) {
                if (lv < lvTop || lv >= nK) continue;
                const uint32_t f = F[i][lv];
                if (f > fm) fm = f;
                uint32_t rank = 0;                                // events flushed before this one: smaller F, or equal F and smaller k
#pragma unroll
                for (int l2 = 0; l2 < NL; ++l2)
                    if (l2 >= lvTop && l2 < nK && l2 != lv && (F[i][l2] < f || (F[i][l2] == f && l2 > lv))) ++rank;
                ord[i] |= (unsigned __int128)(uint32_t)lv << (RT::OBITS * rank);
            }
        }
        fmax[i] = fm;
        w2[i] = (uint32_t)d[i] | (RW == 8 ? ((uint32_t)ord[i] << 5) : 0u);
        uint32_t n = 0, nlev = 0;                                 // nlev: |T_k| per level, 3 bits each, saturating at 7 (RW = 8)
        unsigned long long cnt8 = 0;                              // |T_k| per level, 8 bits each (a long list: saturated at 255)
        bool split = false;                                       // a taxon may own several segments (an entry continues an earlier one of its taxon)
        bool done = true;
        if (!fol[i]) done = walk_segments<Meta>(rp[i], d[i], kLow, getMeta, getTax, nIdx, [&](uint32_t s) {
#pragma unroll
            for (int q = 0; q < INL; ++q) if (n == (uint32_t)q) seg[i][q] = s;
            if constexpr (RW == 8) {
                if (n >= (uint32_t)(INL - 1)) {                       // may end up in the pool (it does when there are more than INL)
                    const uint32_t at = atomicAdd(&sOvfN, 1u);
                    if (at < (uint32_t)GOVF) sOvf[at] = make_uint2(s, (uint32_t)(t * GITEMS + i) | ((n - (uint32_t)(INL - 1)) << 10));
                }
                const int lvLo = kHigh - (int)(s >> 27), lvHi = kHigh - (int)((s >> 22) & 31u);   // + 1 in the fields of its levels
                const unsigned long long upTo = lvHi >= 7 ? ~0ull : ((1ull << (8 * (lvHi + 1))) - 1ull);
                cnt8 += upTo & ~((1ull << (8 * lvLo)) - 1ull) & 0x0101010101010101ull;
            }
            ++n;
            if ((int)((s >> 22) & 31u) > kLow) split = true;
        }, (RW == 8 && COOP) ? (((uint32_t)flags >> 9) & 127u ? ((uint32_t)flags >> 9) & 127u : LONG_STEPS) : 0xFFFFFFFFu);   // (flags bits 9-15: another cut, KASA_LONG_STEPS)
        if (!done) { isLong[i] = true; n = 0; cnt8 = 0; split = false; }   // a LONG list: counted by the whole wavefront below (what was parked is never looked at)
        if (RW == 8 && !COOP && n >= 255u) atomicOr(needCoop, 1u);           // this form cannot count it: the batch is grouped again
        if constexpr (RW == 8) {
#pragma unroll
            for (int lv = 0; lv < 8; ++lv) { const uint32_t cl = (uint32_t)(cnt8 >> (8 * lv)) & 255u; nlev |= (cl < 7u ? cl : 7u) << (3 * lv); }
            cnt8x[i] = cnt8;
        }
        nseg[i] = n;
        w3[i] = RW == 8 ? ((n < 255u ? n : 255u) | (nlev << 8)) : n;
        if (split) w2[i] |= REC_SPLIT;                           // (wide records: bit 29 of word [2] is free as well)
        if (RW == 8 && (nlev & (nlev >> 1) & (nlev >> 2) & 0x249249u)) w2[i] |= REC_SAT;
        if (n > (uint32_t)INL) need += n - (uint32_t)(INL - 1) + 1u + ((w2[i] & REC_SAT) ? POOL_SIZES : 0u);   // pool block: {nseg, [sizes], segments INL-1 ...}
        if (coverage) {                                           // Compare.hpp:926-927: once per matched group, by its first query
            const Key q = qKmer[base + i];
            const int ql = (base + i == 0) ? 0 : lcp_letters<Key>(qKmer[base + i - 1], q);
            walk_segments<Meta>(rp[i], d[i], kLow, getMeta, getTax, nIdx, [&](uint32_t s) {
                for (int k = (int)((s >> 22) & 31u); k <= (int)(s >> 27); ++k)
                    if (ql < group_letters(k)) atomicAdd((unsigned long long *)&cntTotal[(size_t)(kHigh - k) * nTaxa + (s & SEG_TAX_MASK)], 1ull);
            });
        }
    }
    // (cooperative form: the hits of the groups a query heads are needed by the long lists' second pass already -- it counts
    // their profile keys while it places their segments, which saves a sweep over every long list -- so they are made here
    // and parked in LDS; the key phase at the end makes them again for everybody)
    uint32_t longKeysPre[GITEMS] = {0u, 0u};
    if constexpr (RW == 8 && COOP) {
        if (!(flags & 8)) {
            unsigned long long gqE[GITEMS] = {0ull, 0ull};
            const unsigned long long beyondE = lane == 63 ? 0ull : (~0ull << (lane + 1)), fromMeE = ~0ull << lane;
#pragma unroll
            for (int lv = 0; lv < NL; ++lv) {
                if (lv >= nK) continue;
                const int k = kHigh - lv;
                const bool mem0 = d[0] >= k, mem1 = d[1] >= k;
                const bool head0 = mem0 && (((sp[0] >> lv) & 1u) || lane == 0), head1 = mem1 && ((sp[1] >> lv) & 1u);
                const unsigned long long stop0 = __ballot(head0 || !mem0), stop1 = __ballot(head1 || !mem1);
                const unsigned long long s0 = stop0 & beyondE, s1f = stop1 & fromMeE, s1b = stop1 & beyondE;
                const uint32_t e0 = s0 ? 2u * (uint32_t)(__ffsll((long long)s0) - 1) : 128u;
                const uint32_t e1f = s1f ? 2u * (uint32_t)(__ffsll((long long)s1f) - 1) + 1u : 128u;
                const uint32_t e1b = s1b ? 2u * (uint32_t)(__ffsll((long long)s1b) - 1) + 1u : 128u;
                if (head0) gqE[0] |= (unsigned long long)(min(e0, e1f) - 2u * (uint32_t)lane) << (8 * lv);
                if (head1) gqE[1] |= (unsigned long long)(min(e0, e1b) - (2u * (uint32_t)lane + 1u)) << (8 * lv);
            }
            sHitsQ[t * GITEMS + 0] = gqE[0]; sHitsQ[t * GITEMS + 1] = gqE[1];
        }
    }
    // ---- long lists, pass 1 (narrow records): every lane of the wavefront takes entries of the list.  A segment's place in the
    // list is given by its class -- letters in common with the query, descending; left of j before right; nearest first -- so
    // counting the segments per class is enough to place them later (pass 2, once the pool block is allocated).  Here: the
    // number of segments, |T_k| per level (marks at the ends of the level ranges, running sum), the split flag.
    constexpr int LMc = sizeof(Meta) == 1 ? 15 : 255, DSc = sizeof(Meta) == 1 ? 4 : 8;
    auto segOfV = [&](bool ok, uint32_t m, uint32_t tx, int v, int dd, uint32_t &sg) -> bool {   // the segment an entry {meta m, taxon tx} yields, seen from a query of depth dd
        if (!ok) return false;
        const int dup = (int)(m >> DSc);
        const int kFirst = dup < RANGE_LETTERS ? kLow : (dup + 1 > kLow ? dup + 1 : kLow);
        const int kLast = v < dd ? v : dd;
        sg = tx | ((uint32_t)kFirst << 22) | ((uint32_t)kLast << 27);
        return kFirst <= kLast;
    };
    auto segOf = [&](bool ok, uint32_t idx, int v, int dd, uint32_t &sg) -> bool {   // ... entry idx
        if (!ok) return false;
        return segOfV(true, (uint32_t)meta[idx], tax[idx], v, dd, sg);
    };
    // counts of one long list into the wavefront's scratch; returns whether entry j itself yields a segment
    auto coopCount = [&](uint32_t j, int dd, uint32_t &selfSeg, bool &anySplit) -> bool {
        if constexpr (!(RW == 8 && COOP)) return false;               // (the scratch exists in the cooperative form only)
        constexpr int CH = (RW == 8 && COOP) ? 32 : 0;
        uint32_t *cM = &sCo[wv][0], *cH = &sCo[wv][CH];
        if (lane < 32) cM[lane] = 0u;
        cH[lane] = 0u;
        LDS_WAVE_SYNC_G();
        const int chain0 = dd > RANGE_LETTERS ? dd : RANGE_LETTERS;
        const bool selfEmits = segOf(true, j, dd, dd, selfSeg);
        anySplit = selfEmits && (int)((selfSeg >> 22) & 31u) > kLow;
        if (selfEmits && lane == 0) { atomicAdd(&cM[kHigh - (int)(selfSeg >> 27)], 1u); atomicSub(&cM[kHigh - (int)((selfSeg >> 22) & 31u) + 1], 1u); }
        coop_walk<Meta>(meta, tax, nIdx, j, dd, kLow, lane, [&](bool ok, uint32_t, int v, int side, uint32_t em, uint32_t et) {
            uint32_t sg = 0;
            const bool emits = segOfV(ok, em, et, v, dd, sg);
            if (emits) {
                atomicAdd(&cH[(chain0 - v) * 2 + side], 1u);
                atomicAdd(&cM[kHigh - (int)(sg >> 27)], 1u);
                atomicSub(&cM[kHigh - (int)((sg >> 22) & 31u) + 1], 1u);
            }
            if (__ballot(emits && (int)((sg >> 22) & 31u) > kLow) != 0ull) anySplit = true;
        });
        LDS_WAVE_SYNC_G();
        return selfEmits;
    };
    if constexpr (RW == 8 && COOP) {
#pragma unroll
        for (int i = 0; i < GITEMS; ++i) {
            unsigned long long todo = __ballot(isLong[i]);
            while (todo) {                                           // (uniform)
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const uint32_t j = (uint32_t)__shfl((int)rp[i], src);
                const int dd = __shfl(d[i], src);
                uint32_t selfSeg = 0; bool anySplit = false;
                const bool selfEmits = coopCount(j, dd, selfSeg, anySplit);
                const uint32_t n = wave_total(wave_incl_sum(sCo[wv][32 + lane])) + (selfEmits ? 1u : 0u);
                const uint32_t sz = wave_incl_sum(lane < 32 ? sCo[wv][lane] : 0u);     // lane lv: |T| of level lv
                unsigned long long c8 = 0; uint32_t nl = 0;
#define KASA_LV(lv) { const uint32_t c = lane_value<lv>(sz); c8 |= (unsigned long long)(c < 255u ? c : 255u) << (8 * lv); nl |= (c < 7u ? c : 7u) << (3 * lv); }
                KASA_LV(0) KASA_LV(1) KASA_LV(2) KASA_LV(3) KASA_LV(4) KASA_LV(5) KASA_LV(6) KASA_LV(7)
#undef KASA_LV
                if (lane == src) {
                    nseg[i] = n; cnt8x[i] = c8;
                    w3[i] = (n < 255u ? n : 255u) | (nl << 8);
                    if (anySplit) w2[i] |= REC_SPLIT;
                    if (n > (uint32_t)INL) { w2[i] |= REC_SAT; need += n - (uint32_t)(INL - 1) + 1u + POOL_SIZES; }   // (a long list always carries its exact sizes)
                }
                LDS_WAVE_SYNC_G();
            }
        }
    }
    bool poolOk = true;                                              // the workgroup's pool block was allocated (else the host grows the pool and reruns)
    bool parked = false;                                             // the lists' further segments all wait in sOvf (no query with 255 or more, no overflow of the buffer)
    if (__syncthreads_or(need != 0u)) {                              // uniform across the workgroup
        uint32_t total = 0;
        uint32_t off = block_excl_prefix_sum(need, shU, total);
        if (t == 0) {                                                // 64-bit cursor: the host sees how much was asked for, even beyond 2^32
            const unsigned long long at = atomicAdd(poolCursor, (unsigned long long)total);
            sBase = at + total <= (unsigned long long)poolCap ? (uint32_t)at : NOPOS;
        }
        __syncthreads();
        const bool fits = sBase != NOPOS;
        poolOk = fits;
        off += fits ? sBase : 0u;
        // fast way (narrow records): headers by the owners, the waiting segments by everybody.  Not when the buffer
        // overflowed or a query has 255 or more segments (8-bit counts): then the lists are walked again, below.
        bool again = true;
        if constexpr (RW == 8) {
            bool mineOk = true;
#pragma unroll
            for (int i = 0; i < GITEMS; ++i) if (nseg[i] >= 255u || isLong[i]) mineOk = false;
            again = __syncthreads_or((!mineOk || sOvfN > (uint32_t)GOVF) ? 1 : 0) != 0;
            parked = !again;
            if (!again) {
#pragma unroll
                for (int i = 0; i < GITEMS; ++i) {
                    uint32_t dst = NOPOS;
                    if (nseg[i] > (uint32_t)INL) {
                        const bool sat = (w2[i] & REC_SAT) != 0u;
                        if (fits) {
                            pool[off] = nseg[i];
                            if (sat) {
                                uint32_t wds[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    wds[q] = ((uint32_t)(cnt8x[i] >> (16 * q)) & 255u) | (((uint32_t)(cnt8x[i] >> (16 * q + 8)) & 255u) << 16);
                                pool[off + 1] = wds[0]; pool[off + 2] = wds[1]; pool[off + 3] = wds[2]; pool[off + 4] = wds[3];
                            }
                            dst = off + 1u + (sat ? POOL_SIZES : 0u);
                        }
                        seg[i][INL - 1] = off;
                        off += nseg[i] - (uint32_t)(INL - 1) + 1u + (sat ? POOL_SIZES : 0u);
                    }
                    sOvfDst[t * GITEMS + i] = dst;
                }
                __syncthreads();
                const uint32_t nOvf = sOvfN;
                for (uint32_t x = t; x < nOvf; x += GTHREADS) {
                    const uint2 e = sOvf[x];
                    const uint32_t dst = sOvfDst[e.y & 1023u];
                    if (dst != NOPOS) pool[dst + (e.y >> 10)] = e.x;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GITEMS; ++i)
            if (again && nseg[i] > (uint32_t)INL) {
                const bool sat = RW == 8 && (w2[i] & REC_SAT) != 0u;
                if (fits) {                                          // (else: the host grows the pool and reruns)
                    pool[off] = nseg[i];
                    uint32_t w = off + 1 + (sat ? POOL_SIZES : 0u), idx = 0;
                    unsigned long long cA = 0, cB = 0;               // exact |T| of the levels 0..3 and 4..7, 16 bits each
                    if (!isLong[i]) walk_segments<Meta>(rp[i], d[i], kLow, getMeta, getTax, nIdx, [&](uint32_t s) {
                        if (idx >= (uint32_t)(INL - 1)) pool[w++] = s;
                        ++idx;
                        if (sat)
                            for (int lv = kHigh - (int)(s >> 27); lv <= kHigh - (int)((s >> 22) & 31u); ++lv) {
                                if (lv < 4) cA += 1ull << (16 * lv); else cB += 1ull << (16 * (lv - 4));
                            }
                    });
                    if (sat && !isLong[i]) { pool[off + 1] = (uint32_t)cA; pool[off + 2] = (uint32_t)(cA >> 32); pool[off + 3] = (uint32_t)cB; pool[off + 4] = (uint32_t)(cB >> 32); }
                }
                seg[i][INL - 1] = off;
                off += nseg[i] - (uint32_t)(INL - 1) + 1u + (sat ? POOL_SIZES : 0u);
            }
    }
    // ---- long lists, pass 2: the segments to their places -- the first INL (or INL - 1) into the record, the rest into the pool
    // block: place = first place of the segment's class (running sum over the classes) + segments of the class met so far.
    if constexpr (RW == 8 && COOP) {
#pragma unroll
        for (int i = 0; i < GITEMS; ++i) {
            unsigned long long todo = __ballot(isLong[i]);
            while (todo) {                                           // (uniform)
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const uint32_t j = (uint32_t)__shfl((int)rp[i], src);
                const int dd = __shfl(d[i], src);
                const uint32_t n = (uint32_t)__shfl((int)nseg[i], src), off = (uint32_t)__shfl((int)seg[i][INL - 1], src);
                const bool toPool = n > (uint32_t)INL && poolOk;
                const int chain0 = dd > RANGE_LETTERS ? dd : RANGE_LETTERS;
                uint32_t selfSeg = 0; bool anySplit = false;
                const bool selfEmits = coopCount(j, dd, selfSeg, anySplit);
                uint32_t *cH = &sCo[wv][32], *cO = &sCo[wv][96];
                uint32_t Vl = 0, Bl = 0, mineK = 0;                   // the list's profile keys, counted in this sweep
                {
                    const uint32_t cnt = cH[lane];
                    cO[lane] = wave_incl_sum(cnt) - cnt + (selfEmits ? 1u : 0u);       // first place of class `lane`
                    const uint32_t sz = wave_incl_sum(lane < 32 ? sCo[wv][lane] : 0u);
                    const uint32_t s0 = lane_value<0>(sz), s1 = lane_value<1>(sz), s2 = lane_value<2>(sz), s3 = lane_value<3>(sz);
                    const uint32_t s4 = lane_value<4>(sz), s5 = lane_value<5>(sz), s6 = lane_value<6>(sz), s7 = lane_value<7>(sz);
                    auto f16 = [](uint32_t x) -> uint32_t { return x < 0xFFFFu ? x : 0xFFFFu; };
                    if (toPool && lane == 0) {
                        pool[off] = n;
                        pool[off + 1] = f16(s0) | (f16(s1) << 16); pool[off + 2] = f16(s2) | (f16(s3) << 16);
                        pool[off + 3] = f16(s4) | (f16(s5) << 16); pool[off + 4] = f16(s6) | (f16(s7) << 16);
                    }
                    if (selfEmits && lane == 0) sInl[wv][0] = selfSeg;
                    // the levels with hits and the starts of the runs of equal (|T|, hits), as the key phase will see them
                    const unsigned long long hqL = (flags & 8) ? 0ull : sHitsQ[(wv * 64 + src) * GITEMS + i];
                    const uint32_t szl[8] = {f16(s0), f16(s1), f16(s2), f16(s3), f16(s4), f16(s5), f16(s6), f16(s7)};
                    uint32_t pn = 0, pg = 0;
#pragma unroll
                    for (int lv = 0; lv < NL; ++lv) {
                        const uint32_t h = (uint32_t)(hqL >> (8 * lv)) & 255u;
                        if (h) { Vl |= 1u << lv; if (!(pg == h && pn == szl[lv])) Bl |= 1u << lv; }
                        pn = szl[lv]; pg = h;
                    }
                    if (selfEmits && lane == 0 && Vl) mineK += (uint32_t)__popc(((Bl & seg_level_mask(selfSeg, kHigh)) | (Vl & seg_level_mask(selfSeg, kHigh) & (0u - seg_level_mask(selfSeg, kHigh)))));
                }
                LDS_WAVE_SYNC_G();
                coop_walk<Meta>(meta, tax, nIdx, j, dd, kLow, lane, [&](bool ok, uint32_t, int v, int side, uint32_t em, uint32_t et) {
                    uint32_t sg = 0;
                    const bool emits = segOfV(ok, em, et, v, dd, sg);
                    if (emits && Vl) { const uint32_t M = seg_level_mask(sg, kHigh); mineK += (uint32_t)__popc((Bl & M) | (Vl & M & (0u - M))); }
                    const uint32_t c = emits ? (uint32_t)((chain0 - v) * 2 + side) : 0xFFFFu;
                    uint32_t place = 0;
                    unsigned long long rem = __ballot(emits);
                    while (rem) {                                    // class by class (two or three per chunk): places in lane order
                        const int l0 = __ffsll((long long)rem) - 1;
                        const uint32_t c0 = (uint32_t)__shfl((int)c, l0);
                        const unsigned long long m = __ballot(c == c0);
                        if (c == c0) place = cO[c0] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        LDS_WAVE_SYNC_G();
                        if (lane == l0) cO[c0] += (uint32_t)__popcll(m);
                        LDS_WAVE_SYNC_G();
                        rem &= ~m;
                    }
                    if (emits) {
                        if (place < (uint32_t)INL) sInl[wv][place] = sg;
                        if (toPool && place >= (uint32_t)(INL - 1)) pool[off + 1u + POOL_SIZES + place - (uint32_t)(INL - 1)] = sg;
                    }
                });
                LDS_WAVE_SYNC_G();
                const uint32_t keysOfList = wave_total(wave_incl_sum(mineK));
                if (lane == src) {
                    const uint32_t inl = n <= (uint32_t)INL ? n : (uint32_t)(INL - 1);
#pragma unroll
                    for (int q = 0; q < INL; ++q) if ((uint32_t)q < inl) seg[i][q] = sInl[wv][q];
                    longKeysPre[i] = keysOfList;
                }
                LDS_WAVE_SYNC_G();
            }
        }
    }
    // ---- followers take their leader's words (all lanes run the shuffles)
    {
        const unsigned long long l0 = lead0, l1 = lead1;
        const unsigned long long before = (l0 | l1) & ((1ull << lane) - 1ull);       // leaders in the lanes before (lane 0: none, and it never follows)
        const int src = fol[0] ? 63 - __clzll((long long)before) : lane;             // the nearest one: item 1 of that lane if it leads, else item 0
        const bool srcItem1 = ((l1 >> src) & 1ull) != 0ull;
        auto take = [&](uint32_t a, uint32_t b) -> uint32_t {
            const uint32_t va = (uint32_t)__shfl((int)a, src), vb = (uint32_t)__shfl((int)b, src);
            return srcItem1 ? vb : va;
        };
        const uint32_t FL = REC_SPLIT | (RW == 8 ? REC_SAT : 0u);
        const uint32_t g3 = take(w3[0], w3[1]), g2 = take(w2[0] & FL, w2[1] & FL);
        uint32_t gs[INL];
#pragma unroll
        for (int q = 0; q < INL; ++q) gs[q] = take(seg[0][q], seg[1][q]);
        if (fol[0]) {
            w3[0] = g3; w2[0] |= g2;
#pragma unroll
            for (int q = 0; q < INL; ++q) seg[0][q] = gs[q];
        }
        if (fol[1]) {
            w3[1] = w3[0]; w2[1] |= w2[0] & FL;
#pragma unroll
            for (int q = 0; q < INL; ++q) seg[1][q] = seg[0][q];
        }
    }
    if constexpr (RW == 16) __syncthreads();                         // the index span is dead: the output stage takes its place
    // ---- 4. the record
#pragma unroll
    for (int i = 0; i < GITEMS; ++i) {
        const uint32_t p = base + i;
        if (p >= nQ) continue;
        if constexpr (RW == 8) {
            const uint32_t slot = slotOf ? slotOf[p] : p;
            uint4 *o = reinterpret_cast<uint4 *>(rec + (size_t)slot * cellW);   // (cellW = 16: the few tiles group2_kernel lists, in its cells)
            o[0] = make_uint4(p, fmax[i], w2[i], w3[i]);
            o[1] = make_uint4(seg[i][0], seg[i][1], seg[i][2], seg[i][3]);
        }
    }
    if constexpr (RW == 16) {
        uint4 *mine = sOut + wv * 256;
#pragma unroll
        for (int i = 0; i < GITEMS; ++i) {
            const uint32_t p = base + i;
            const uint32_t slot = p < nQ ? (slotOf ? slotOf[p] : p) : NOPOS;
            mine[lane * 4 + 0] = make_uint4(p, fmax[i], w2[i], w3[i]);
            mine[lane * 4 + 1] = make_uint4((uint32_t)ord[i], (uint32_t)(ord[i] >> 32), (uint32_t)(ord[i] >> 64), (uint32_t)(ord[i] >> 96));
            mine[lane * 4 + 2] = make_uint4(seg[i][0], seg[i][1], seg[i][2], seg[i][3]);
            mine[lane * 4 + 3] = make_uint4(seg[i][4], seg[i][5], seg[i][6], seg[i][7]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {                            // instruction k: the records of lanes 16k .. 16k+15, a quad each
                const uint32_t sk = (uint32_t)__shfl((int)slot, 16 * k + (lane >> 2));
                if (sk != NOPOS) reinterpret_cast<uint4 *>(rec + (size_t)sk * RW)[lane & 3] = mine[k * 64 + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- the profile (after the records have left: their stores drain meanwhile, and their registers are free): countAll[k][t] += |H| / |T|, countUnique[k][t] += |H| iff |T| = 1 (Compare.hpp:922-925) is a sum over
    // (group, taxon) -- it is made HERE, from the sorted queries, not per read: the first query a wavefront holds of a level-k
    // group adds the group's hits inside the wavefront (sums are associative: a group that spans wavefronts adds in pieces)
    // to every taxon of T_k.  Such a query is a leader (a follower shares all its levels' groups with its predecessor).  The
    // contributions leave as keys {levels, |T|, taxon, hits} -- one key for a run of levels with the same |T| and hits -- for
    // profile_group_table_kernel (one allocation per workgroup); the per-read side (score_*) carries no profile bookkeeping,
    // and a batch without per-read scores needs no score stage at all.
    if (GpOf<RW>::v && !(flags & 8)) {                               // (flags 8, 16: timing taps -- no key phase; no key stores)
        // hits per level: members of the level's group from this query on, if it is the group's first in the wavefront (else 0)
        unsigned long long gq[GITEMS] = {0ull, 0ull};               // 8 bits per level
        {
            const unsigned long long beyond = lane == 63 ? 0ull : (~0ull << (lane + 1)), fromMe = ~0ull << lane;
#pragma unroll
            for (int lv = 0; lv < NL; ++lv) {
                if (lv >= nK) continue;
                const int k = kHigh - lv;
                const bool mem0 = d[0] >= k, mem1 = d[1] >= k;
                const bool head0 = mem0 && (((sp[0] >> lv) & 1u) || lane == 0), head1 = mem1 && ((sp[1] >> lv) & 1u);
                // where a group that begins here ends: at the next head or the next query that is not matched this deep
                const unsigned long long stop0 = __ballot(head0 || !mem0), stop1 = __ballot(head1 || !mem1);
                const unsigned long long s0 = stop0 & beyond, s1f = stop1 & fromMe, s1b = stop1 & beyond;
                const uint32_t e0 = s0 ? 2u * (uint32_t)(__ffsll((long long)s0) - 1) : 128u;
                const uint32_t e1f = s1f ? 2u * (uint32_t)(__ffsll((long long)s1f) - 1) + 1u : 128u;
                const uint32_t e1b = s1b ? 2u * (uint32_t)(__ffsll((long long)s1b) - 1) + 1u : 128u;
                if (head0) gq[0] |= (unsigned long long)(min(e0, e1f) - 2u * (uint32_t)lane) << (8 * lv);
                if (head1) gq[1] |= (unsigned long long)(min(e0, e1b) - (2u * (uint32_t)lane + 1u)) << (8 * lv);
            }
        }
        // Per item: the levels with hits (V), where a run of levels with the same (|T|, hits) begins (B), and |T| per level in
        // 16-bit fields (the 8-bit fields of cnt8 have wrapped at 255 or more segments: then the exact sizes group_kernel put
        // into the pool block).  A segment's keys: one per run that begins inside its level range -- counted and cut out
        // with a few bit operations; the segments come from the registers and, beyond the inline ones, from the pool block the
        // workgroup has just written (no second walk).
        // Per item: the levels with hits (V), where a run of levels with the same (|T|, hits) begins (B).  A segment's keys: one
        // per run that begins inside its level range -- counted and cut out with a few bit operations.  The segments: the
        // inline ones from their owner's registers; the further ones of a list from where walk 1 parked them (sOvf) -- every
        // thread takes a share of those, with the owner's level data from LDS (keys need no order); only when the tile could
        // not park them (a query with 255 or more, a full buffer) from the pool block, or by walks of their own.
        uint32_t Vm[GITEMS], Bm[GITEMS];
        uint32_t needK = 0;
        auto segAt = [&](int i, uint32_t idx) -> uint32_t {          // idx >= INL - 1 of a list that continues in the pool
            const uint32_t skip = (w2[i] & REC_SAT) ? POOL_SIZES : 0u;
            return pool[seg[i][INL - 1] + 1u + skip + idx - (uint32_t)(INL - 1)];
        };
        auto forSegs = [&](int i, auto f) {                          // the owner's share of item i's segments
            const uint32_t ns = nseg[i];
            const uint32_t inl = ns <= (uint32_t)INL ? ns : (uint32_t)(INL - 1);
#pragma unroll
            for (int q = 0; q < INL; ++q) if ((uint32_t)q < inl) f(seg[i][q]);
            if (!parked) for (uint32_t q = inl; q < ns; ++q) f(segAt(i, q));
        };
        // a long list: |T| per level from the exact sizes in its pool block (8-bit fields saturate), its segments by a walk
        auto longSize = [&](int i, int lv) -> uint32_t { return (pool[seg[i][INL - 1] + 1u + (uint32_t)(lv >> 1)] >> (16 * (lv & 1))) & 0xFFFFu; };
        auto longList = [&](int i) -> bool { return isLong[i] && nseg[i] > (uint32_t)INL; };
        auto startsOf = [](uint32_t M, uint32_t V, uint32_t B) -> uint32_t { return (B & M) | (V & M & (0u - M)); };
        // one key per run: levels lo .. hi, |T| and hits of level lo
        auto cutRuns = [](uint32_t M, uint32_t V, uint32_t B, auto emit) {
            uint32_t starts = (B & M) | (V & M & (0u - M));
            const uint32_t stops = (B | ~V) & M;                     // a run ends before the next run's start, a level without hits, or with the segment
            while (starts) {
                const int lo = __ffs((int)starts) - 1;
                starts &= starts - 1u;
                const uint32_t above = stops & ~((2u << lo) - 1u);
                emit(lo, above ? __ffs((int)above) - 2 : 31 - __clz((int)M));
            }
        };
#pragma unroll
        for (int i = 0; i < GITEMS; ++i) {
            Vm[i] = 0; Bm[i] = 0;
            if (nseg[i] == 0u || gq[i] == 0ull || (nseg[i] > (uint32_t)INL && !poolOk)) gq[i] = 0ull;
            if (gq[i] != 0ull) {
                uint32_t pn = 0, pg = 0;
                const bool lng = longList(i);
#pragma unroll
                for (int lv = 0; lv < NL; ++lv) {
                    const uint32_t n = lng ? longSize(i, lv) : ((uint32_t)(cnt8x[i] >> (8 * lv)) & 255u);
                    const uint32_t h = (uint32_t)(gq[i] >> (8 * lv)) & 255u;
                    if (h) { Vm[i] |= 1u << lv; if (!(pg == h && pn == n)) Bm[i] |= 1u << lv; }
                    pn = n; pg = h;                                   // (pg = 0 after a level without hits: the next one begins a run)
                }
            }
            sHitsQ[t * GITEMS + i] = gq[i]; sSizeQ[t * GITEMS + i] = cnt8x[i]; sRunQ[t * GITEMS + i] = (uint16_t)(Vm[i] | (Bm[i] << 8));
            if (gq[i] == 0ull) continue;
            if (longList(i)) continue;                                // (counted by the whole wavefront, below)
            forSegs(i, [&](uint32_t sg) { needK += (uint32_t)__popc(startsOf(seg_level_mask(sg, kHigh), Vm[i], Bm[i])); });
        }
        // long lists: the keys of their segments were counted by the second pass (longKeysPre); they are written, further down,
        // by all lanes of the wavefront
        uint32_t longKeys[GITEMS] = {0u, 0u};
        if constexpr (COOP) {
#pragma unroll
            for (int i = 0; i < GITEMS; ++i)
                if (gq[i] != 0ull && longList(i)) { longKeys[i] = longKeysPre[i]; needK += longKeysPre[i]; }
        }
        __syncthreads();                                             // the owners' level data, the pool blocks (other threads have filled them)
        const uint32_t nPark = (parked && poolOk) ? sOvfN : 0u;
        for (uint32_t x = t; x < nPark; x += GTHREADS) {
            const uint2 e = sOvf[x];
            const uint32_t o = e.y & 1023u;
            if (sOvfDst[o] == NOPOS) continue;                       // (its query has no list after all: the segment is inline)
            const uint32_t vb = sRunQ[o];
            needK += (uint32_t)__popc(startsOf(seg_level_mask(e.x, kHigh), vb & 255u, vb >> 8));
        }
        if (__syncthreads_or(needK != 0u)) {
            uint32_t totalK = 0;
            uint32_t kw = block_excl_prefix_sum(needK, shU, totalK);
            if (t == 64) {
                const unsigned long long at = atomicAdd(keyCursor, (unsigned long long)totalK);
                sBaseK = at + totalK <= (unsigned long long)keyCap ? (uint32_t)at : NOPOS;
            }
            __syncthreads();
            // The workgroup's keys are one piece of the key buffer.  A thread's keys leave from inside its loops, a few lanes at a
            // time and with gaps between the lanes' places (partial-cell writes); so, when the index span in LDS is dead (no walks
            // in this phase) and the keys fit there, they are collected in LDS and copied out in whole lines.
            constexpr uint32_t KSTAGE = (uint32_t)(SPAN_BYTES / 8);
            const bool staged = parked && totalK <= KSTAGE && sBaseK != NOPOS;
            unsigned long long *sKeys = reinterpret_cast<unsigned long long *>(sRaw);
            if (COOP && sBaseK != NOPOS) {                              // (uniform) the long lists' keys, by all lanes: a leader's keys begin with them
#pragma unroll
                for (int i = 0; i < GITEMS; ++i) {
                    unsigned long long todo = __ballot(longKeys[i] != 0u);
                    uint32_t before = 0;                                // keys of the leader's long item 0 (they come first)
                    if (i == 1) before = longKeys[0];
                    while (todo) {
                        const int src = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        const uint32_t j = (uint32_t)__shfl((int)rp[i], src), V = (uint32_t)__shfl((int)Vm[i], src), B = (uint32_t)__shfl((int)Bm[i], src);
                        const int dd = __shfl(d[i], src);
                        const uint32_t off = (uint32_t)__shfl((int)seg[i][INL - 1], src);
                        const uint32_t hLo = (uint32_t)__shfl((int)(uint32_t)gq[i], src), hHi = (uint32_t)__shfl((int)(uint32_t)(gq[i] >> 32), src);
                        const unsigned long long hq = ((unsigned long long)hHi << 32) | hLo;
                        uint32_t at = sBaseK + (uint32_t)__shfl((int)(kw + before), src);
                        auto putL = [&](uint32_t sg, uint32_t &w) {
                            cutRuns(seg_level_mask(sg, kHigh), V, B, [&](int lo, int hi) {
                                const uint32_t n = (pool[off + 1u + (uint32_t)(lo >> 1)] >> (16 * (lo & 1))) & 0xFFFFu, hits = (uint32_t)(hq >> (8 * lo)) & 255u, tx = sg & SEG_TAX_MASK;
                                if (flags & 16) { ++w; return; }
                                unsigned long long key = 0ull;
                                if (n < (1u << PL.nb)) key = group_key((uint32_t)lo, (uint32_t)hi, n, tx, hits);
                                else for (int lv = lo; lv <= hi; ++lv) fixed_add(cntAllHi, cntAllMid, cntAllLo, (size_t)lv * nTaxa + tx, hits, n);
                                profKeys[w++] = key;
                            });
                        };
                        uint32_t selfSeg = 0;
                        if (segOf(true, j, dd, dd, selfSeg)) {          // entry j's own segment: lane 0
                            const uint32_t c = (uint32_t)__popc(startsOf(seg_level_mask(selfSeg, kHigh), V, B));
                            if (lane == 0) { uint32_t w = at; putL(selfSeg, w); }
                            at += c;
                        }
                        coop_walk<Meta>(meta, tax, nIdx, j, dd, kLow, lane, [&](bool ok, uint32_t, int v, int, uint32_t em, uint32_t et) {
                            uint32_t sg = 0;
                            const bool emits = segOfV(ok, em, et, v, dd, sg);
                            const uint32_t c = emits ? (uint32_t)__popc(startsOf(seg_level_mask(sg, kHigh), V, B)) : 0u;
                            const uint32_t incl = wave_incl_sum(c);
                            if (c) { uint32_t w = at + incl - c; putL(sg, w); }
                            at += wave_total(incl);
                        });
                    }
                }
            }
            if (sBaseK != NOPOS && needK) {
                if (!staged) kw += sBaseK;
                kw += longKeys[0] + longKeys[1];
                auto put = [&](int lo, int hi, uint32_t n, uint32_t hits, uint32_t tx) {
                    if (flags & 16) { ++kw; return; }
                    unsigned long long key = 0ull;                      // (hits = 0: skipped by the table kernel)
                    if (n < (1u << PL.nb)) key = group_key((uint32_t)lo, (uint32_t)hi, n, tx, hits);
                    else                                                // a set too large for the key's field: straight to the tables
                        for (int lv = lo; lv <= hi; ++lv) fixed_add(cntAllHi, cntAllMid, cntAllLo, (size_t)lv * nTaxa + tx, hits, n);
                    if (staged) sKeys[kw++] = key; else profKeys[kw++] = key;
                };
#pragma unroll
                for (int i = 0; i < GITEMS; ++i) {
                    if (gq[i] == 0ull || longList(i)) continue;
                    forSegs(i, [&](uint32_t sg) {
                        cutRuns(seg_level_mask(sg, kHigh), Vm[i], Bm[i], [&](int lo, int hi) {
                            put(lo, hi, (uint32_t)(cnt8x[i] >> (8 * lo)) & 255u, (uint32_t)(gq[i] >> (8 * lo)) & 255u, sg & SEG_TAX_MASK);
                        });
                    });
                }
                for (uint32_t x = t; x < nPark; x += GTHREADS) {
                    const uint2 e = sOvf[x];
                    const uint32_t o = e.y & 1023u;
                    if (sOvfDst[o] == NOPOS) continue;
                    const uint32_t vb = sRunQ[o];
                    const unsigned long long hq = sHitsQ[o], sq = sSizeQ[o];
                    cutRuns(seg_level_mask(e.x, kHigh), vb & 255u, vb >> 8, [&](int lo, int hi) {
                        put(lo, hi, (uint32_t)(sq >> (8 * lo)) & 255u, (uint32_t)(hq >> (8 * lo)) & 255u, e.x & SEG_TAX_MASK);
                    });
                }
            }
            if (staged && !(flags & 16)) {                              // (uniform)
                __syncthreads();
                for (uint32_t x = t; x < totalK; x += GTHREADS) profKeys[(size_t)sBaseK + x] = sKeys[x];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// group2: the group stage of ordinary tiles (narrow records), leaders compacted onto lanes
// ------------------------------------------------------------------------------------------------
// group_kernel gives a thread two sorted queries and lets every lane do everything for its own queries.  But 60 % of the
// queries are FOLLOWERS (their (representative, depth) is their predecessor's): lanes that hold them idle in lockstep while
// their neighbours walk the index and cut profile keys -- the two loops that are most of the kernel's instructions -- and the
// kernel, which ends in a scatter the chip could retire in 48 ms, took 71 issuing them (round 4: 45e9 wavefront
// instructions).  Here the work is dealt out twice:
//   A  (sorted layout, two queries per thread)  flush positions, order of the events, hits of the groups a query heads --
//      the ballot work over neighbours in sorted order.  Leaders are numbered through the tile and leave {rep, depth, hits} in LDS.
//   B  (ONE LEADER PER LANE, dense: ~410 of a tile's 1024 queries)  the walk over the index span in LDS: segments straight
//      into LDS (the first four of a list; further ones parked), level sizes as a histogram of the segments' last levels
//      (|T_k| = its running sum: a multiplication), the leader's own profile keys counted.
//   C  ONE allocation for pool block and keys (a single running sum over the workgroup, two atomics).
//   D  (sorted layout)  every query takes its leader's words from LDS and stores its record.
//   E  (leaders, dense)  the profile keys, staged in the dead index span and written in whole lines.
// What does not fit this scheme is not handled here at all: a tile with a walk that leaves the staged span, runs beyond
// G2_STEPS entries (a conserved k-mer: the cooperative walk's business) or parks more than G2_OVF segments puts its number on
// a list before it has written anything, and the host gives the listed tiles to group_kernel<COOP> (tileList).  Semantics as
// group_kernel: Compare.hpp:747-766 (duplicates join their group), :841-853, :917-955 (markTaxIDs), :922-925 (profile).
static constexpr int G2_SPAN = 1792;              // index entries staged per tile (the tile's representatives +- GMARGIN: ~520 at C2)
static constexpr int G2_OVF = 1024;               // parked segments per tile (the default level count; 512 with up to eight levels: LDS)
static constexpr uint32_t G2_STEPS = 40;          // entries one leader's walk may visit (so that n < 255: 8-bit level sizes)

// two running sums over the workgroup at once (one barrier); sh: 2 * GTHREADS / 64 words
__device__ __forceinline__ void block_excl_prefix_sum2(uint32_t a, uint32_t b, uint32_t *sh, uint32_t &offA, uint32_t &offB, uint32_t &totA, uint32_t &totB)
{
    constexpr int NW = GTHREADS / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ia = wave_incl_sum(a), ib = wave_incl_sum(b);
    if (lane == 63) { sh[wv] = ia; sh[NW + wv] = ib; }
    __syncthreads();
    uint32_t ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const uint32_t x = sh[w], y = sh[NW + w]; if (w < wv) { ba += x; bb += y; } ta += x; tb += y; }
    offA = ba + ia - a; offB = bb + ib - b; totA = ta; totB = tb;
}

template <class Key, int NKT, int VAR = 0, int PARKX = 1>             // VAR: timing variants (KASA_G2_VAR), the product is 0; PARKX: a park buffer that many times as large
__device__ __forceinline__ void group2_tile(const uint32_t tile,
    const Key *__restrict__ qKmer, const uint8_t *__restrict__ depth, const uint32_t *__restrict__ rep,
    const uint32_t *__restrict__ slotOf, uint32_t nQ, const uint32_t *__restrict__ tileNext, uint32_t nTiles,
    const typename KeyTraits<Key>::Meta *__restrict__ meta, const uint32_t *__restrict__ tax, uint32_t nIdx, int kHigh, int kLow,
    uint32_t *__restrict__ rec, uint32_t *__restrict__ pool, uint32_t poolCap, unsigned long long *__restrict__ poolCursor, int flags,
    uint32_t nTaxa, uint64_t *__restrict__ profKeys, uint32_t keyCap, unsigned long long *__restrict__ keyCursor, ProfLayout PL,
    uint64_t *__restrict__ cntAllHi, uint64_t *__restrict__ cntAllMid, uint64_t *__restrict__ cntAllLo,
    uint32_t *__restrict__ slowCount, uint32_t *__restrict__ slowList, uint32_t cellW)
{
    typedef RecTraits<8> RT;
    constexpr int NL = NKT ? NKT : RT::LEVELS, INL = RT::INL, NW = GTHREADS / 64;
    typedef typename KeyTraits<Key>::Meta Meta;
    constexpr int LM = sizeof(Meta) == 1 ? 15 : 255, DS = sizeof(Meta) == 1 ? 4 : 8;
    constexpr int SPAN_BYTES = G2_SPAN * 4 + (G2_SPAN + 8) * (int)sizeof(Meta);      // (+ the sentinel entry behind the span)
    __shared__ __attribute__((aligned(16))) unsigned char sRaw[SPAN_BYTES];
    uint32_t *sTax = reinterpret_cast<uint32_t *>(sRaw);
    Meta *sMeta = reinterpret_cast<Meta *>(sRaw + G2_SPAN * 4);
    // per LEADER of the tile (numbered in sorted order):
    __shared__ uint32_t sA[TILE];                       // its representative index entry; from phase B on word [3] of its record
    __shared__ uint32_t sDF[TILE];                      // depth | flags << 8 (1 split, 2 saturated) | levels with hits << 16 | run starts << 24
    __shared__ uint4 sSeg[TILE];                        // the first four segments; [3] = the pool block of a longer list
    // hits of the groups it heads inside its wavefront and |T| per level, 8 bits per level each: two 64-bit words, or -- six
    // levels, the default -- 48 + 48 bits in three 32-bit words (the room that saves is the park buffer's second half)
    constexpr bool PACK6 = NL <= 6;
    constexpr int OVF = (PACK6 ? G2_OVF : G2_OVF / 2) * PARKX;
    __shared__ uint32_t sHS[(PACK6 ? 3 : 4) * TILE];
    __shared__ uint2 sOvf[OVF];                         // {segment, leader | place in its pool list << 10}: segments beyond the FOURTH
    auto putHits = [&](uint32_t L, unsigned long long h) {
        sHS[L] = (uint32_t)h;
        if constexpr (PACK6) sHS[TILE + L] = (uint32_t)(h >> 32) & 0xFFFFu; else sHS[TILE + L] = (uint32_t)(h >> 32);
    };
    auto getHits = [&](uint32_t L) -> unsigned long long {
        const uint32_t hi = sHS[TILE + L];
        return (unsigned long long)sHS[L] | ((unsigned long long)(PACK6 ? (hi & 0xFFFFu) : hi) << 32);
    };
    auto putSizes = [&](uint32_t L, unsigned long long c8) {       // (after putHits, by the same thread)
        if constexpr (PACK6) { sHS[TILE + L] |= (uint32_t)(c8 & 0xFFFFull) << 16; sHS[2 * TILE + L] = (uint32_t)(c8 >> 16); }
        else { sHS[2 * TILE + L] = (uint32_t)c8; sHS[3 * TILE + L] = (uint32_t)(c8 >> 32); }
    };
    auto getSizes = [&](uint32_t L) -> unsigned long long {
        if constexpr (PACK6) return (unsigned long long)(sHS[TILE + L] >> 16) | ((unsigned long long)sHS[2 * TILE + L] << 16);
        else return (unsigned long long)sHS[2 * TILE + L] | ((unsigned long long)sHS[3 * TILE + L] << 32);
    };
    __shared__ uint32_t shU[2 * NW], sFirst[NW][NL], sRepLo[NW], sRepHi[NW], sLeadN[NW];
    __shared__ uint32_t sBase, sBaseK, sOvfN, sBail;
    if (threadIdx.x == 0) { sOvfN = 0u; sBail = 0u; }
    const int nK = NKT ? NKT : kHigh - kLow + 1;
    const uint32_t allLv = (1u << nK) - 1u;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t base = tile * TILE + (uint32_t)t * 2u;
    // ---- A. sorted layout
    int d[2];
    uint32_t rp[2], sp[2], slot[2];
    {
        const bool in0 = base < nQ, in1 = base + 1u < nQ;
        // (all four arrays carry 64 bytes of slack: the pair loads of the last thread stay inside them)
        const uint32_t dd2 = in0 ? (uint32_t)*reinterpret_cast<const uint16_t *>(depth + base) : 0u;
        const uint2 rr = in0 ? *reinterpret_cast<const uint2 *>(rep + base) : make_uint2(0u, 0u);
        uint2 ss = make_uint2(base, base + 1u);
        if (slotOf && in0) ss = *reinterpret_cast<const uint2 *>(slotOf + base);
        const Key q0 = in0 ? qKmer[base] : (Key)0, q1 = in1 ? qKmer[base + 1u] : (Key)0;
        const int ql0 = (!in0 || base == 0u) ? 0 : lcp_letters<Key>(qKmer[base - 1u], q0);
        const int ql1 = in1 ? lcp_letters<Key>(q0, q1) : 0;
        d[0] = in0 ? (int)(dd2 & 255u) : 0; d[1] = in1 ? (int)(dd2 >> 8) : 0;
        rp[0] = in0 ? rr.x : 0u; rp[1] = in1 ? rr.y : 0u;
        slot[0] = in0 ? ss.x : NOPOS; slot[1] = in1 ? ss.y : NOPOS;
        sp[0] = in0 ? special_mask(ql0, d[0], kHigh, allLv) : 0u;
        sp[1] = in1 ? special_mask(ql1, d[1], kHigh, allLv) : 0u;
    }
    bool fol[2];
    {
        const uint32_t prevRp = (uint32_t)__shfl_up((int)rp[1], 1);
        const int prevD = __shfl_up(d[1], 1);
        fol[0] = lane > 0 && d[0] != 0 && d[0] == prevD && rp[0] == prevRp;
        fol[1] = d[1] != 0 && d[1] == d[0] && rp[1] == rp[0];
    }
    const unsigned long long lead0 = __ballot(d[0] != 0 && !fol[0]), lead1 = __ballot(d[1] != 0 && !fol[1]);
    {   // span of the representatives (nearly monotone in p): first and last matched query of the wavefront
        const unsigned long long m0 = __ballot(d[0] != 0), m1 = __ballot(d[1] != 0);
        const unsigned long long any = m0 | m1;
        uint32_t lo = NOPOS, hi = 0;
        if (any) {
            const int lf = __ffsll((long long)any) - 1, ll = 63 - __clzll((long long)any);
            const uint32_t a0 = __shfl(rp[0], lf), a1 = __shfl(rp[1], lf), b0 = __shfl(rp[0], ll), b1 = __shfl(rp[1], ll);
            lo = ((m0 >> lf) & 1ull) ? a0 : a1;
            hi = ((m1 >> ll) & 1ull) ? b1 : b0;
        }
        if (lane == 0) { sRepLo[wv] = lo; sRepHi[wv] = hi; sLeadN[wv] = (uint32_t)(__popcll(lead0) + __popcll(lead1)); }
    }
    // flush positions F_k(p) (as group_kernel) and, from the same masks, the hits of the groups a query heads
    const unsigned long long above = (lane == 63) ? 0ull : (~0ull << (lane + 1)), fromMe = ~0ull << lane;
    uint32_t F[2][NL];
    uint32_t open0 = 0, open1 = 0;
    unsigned long long gq[2] = {0ull, 0ull};
#pragma unroll
    for (int lv = 0; lv < NL; ++lv) {
        F[0][lv] = NOPOS; F[1][lv] = NOPOS;
        if (lv >= nK) continue;
        const bool s0 = (sp[0] >> lv) & 1u, s1 = (sp[1] >> lv) & 1u;
        const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1);
        const unsigned long long any = b0 | b1;
        uint32_t next = NOPOS;
        const unsigned long long hi = any & above;
        if (hi) {
            const int l2 = __ffsll((long long)hi) - 1;
            next = tile * TILE + (uint32_t)(wv * 64 + l2) * 2u + (((b0 >> l2) & 1ull) ? 0u : 1u);
        }
        F[1][lv] = next;
        F[0][lv] = s1 ? base + 1u : next;
        if (next == NOPOS) { open1 |= 1u << lv; if (!s1) open0 |= 1u << lv; }
        if (lane == 0) {
            uint32_t first = NOPOS;
            if (any) { const int l2 = __ffsll((long long)any) - 1; first = tile * TILE + (uint32_t)(wv * 64 + l2) * 2u + (((b0 >> l2) & 1ull) ? 0u : 1u); }
            sFirst[wv][lv] = first;
        }
        if (!(flags & 8)) {
            // members of the level's group from this query on, if it is the group's first in the wavefront: the group ends at the
            // next head or the next query that is not matched this deep
            const int k = kHigh - lv;
            const bool mem0 = d[0] >= k, mem1 = d[1] >= k;
            const bool head0 = mem0 && (s0 || lane == 0), head1 = mem1 && s1;
            const unsigned long long stop0 = __ballot(head0 || !mem0), stop1 = __ballot(head1 || !mem1);
            const unsigned long long t0 = stop0 & above, t1f = stop1 & fromMe, t1b = stop1 & above;
            const uint32_t e0 = t0 ? 2u * (uint32_t)(__ffsll((long long)t0) - 1) : 128u;
            const uint32_t e1f = t1f ? 2u * (uint32_t)(__ffsll((long long)t1f) - 1) + 1u : 128u;
            const uint32_t e1b = t1b ? 2u * (uint32_t)(__ffsll((long long)t1b) - 1) + 1u : 128u;
            if (head0) gq[0] |= (unsigned long long)(min(e0, e1f) - 2u * (uint32_t)lane) << (8 * lv);
            if (head1) gq[1] |= (unsigned long long)(min(e0, e1b) - (2u * (uint32_t)lane + 1u)) << (8 * lv);
        }
    }
    __syncthreads();
    {   // levels not closed inside the wavefront: the first closing position in the wavefronts behind, else the tiles behind.
        // Lane lv looks that up for level lv (one short loop per wavefront, not one per lane and level); all take it from there.
        uint32_t nx = NOPOS;
        if (lane < nK) {
            for (int w = wv + 1; w < NW; ++w) { const uint32_t o = sFirst[w][lane]; if (o != NOPOS) { nx = o; break; } }
            if (nx == NOPOS) nx = tileNext[(size_t)lane * nTiles + tile];
        }
#pragma unroll
        for (int lv = 0; lv < NL; ++lv) {
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)nx, lv);
            if ((open1 >> lv) & 1u) F[1][lv] = v;
            if ((open0 >> lv) & 1u) F[0][lv] = v;
        }
    }
    uint32_t spanLo = NOPOS, spanHi = 0, lbase = 0, nLead = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        spanLo = min(spanLo, sRepLo[w]); spanHi = max(spanHi, sRepHi[w]);
        const uint32_t o = sLeadN[w];
        if (w < wv) lbase += o;
        nLead += o;
    }
    uint32_t spanN = 0;
    if (spanLo != NOPOS) {
        spanLo = spanLo > GMARGIN ? spanLo - GMARGIN : 0u;
        spanHi = min(spanHi + GMARGIN + 1u, nIdx);
        spanN = min(spanHi - spanLo, (uint32_t)G2_SPAN);
        for (uint32_t x = t; x < spanN; x += GTHREADS) { sTax[x] = tax[spanLo + x]; sMeta[x] = meta[spanLo + x]; }
        if (t == 0) sMeta[spanN] = (Meta)0;                          // the sentinel: shares nothing, ends a walk (which is then reported)
    }
    // the tile's leaders, numbered in sorted order: myL = the last leader at or before the query
    uint32_t myL[2];
    {
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t c0 = (uint32_t)(__popcll(lead0 & below) + __popcll(lead1 & below));
        const uint32_t isL0 = (uint32_t)((lead0 >> lane) & 1ull), isL1 = (uint32_t)((lead1 >> lane) & 1ull);
        myL[0] = lbase + c0 + isL0 - 1u;
        myL[1] = myL[0] + isL1;
        if (isL0) { sA[myL[0]] = rp[0]; putHits(myL[0], gq[0]); sDF[myL[0]] = (uint32_t)d[0]; }
        if (isL1) { sA[myL[1]] = rp[1]; putHits(myL[1], gq[1]); sDF[myL[1]] = (uint32_t)d[1]; }
    }
    // order of the query's events (rank = events flushed before: smaller F, or equal F and smaller k), Fmax
    uint32_t fmax[2], w2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fmax[i] = 0; w2[i] = 0;
        if (d[i] == 0) continue;
        const int lvTop = kHigh - d[i];
        uint32_t rank[NL];
#pragma unroll
        for (int lv = 0; lv < NL; ++lv) rank[lv] = 0;
#pragma unroll
        for (int a = 0; a < NL; ++a)
#pragma unroll
            for (int b2 = a + 1; b2 < NL; ++b2) {
                const bool both = a >= lvTop && b2 < nK;
                const bool aFirst = F[i][a] < F[i][b2];           // a tie goes to b2 (the smaller k)
                rank[b2] += (both && aFirst) ? 1u : 0u;
                rank[a] += (both && !aFirst) ? 1u : 0u;
            }
        uint32_t fm = 0, ord = 0;
#pragma unroll
        for (int lv = 0; lv < NL; ++lv) {
            if (lv < lvTop || lv >= nK) continue;
            const uint32_t f = F[i][lv];
            if (f > fm) fm = f;
            ord |= (uint32_t)lv << (RT::OBITS * rank[lv]);
        }
        fmax[i] = fm;
        w2[i] = (uint32_t)d[i] | (ord << 5);
    }
    __syncthreads();
    if (flags & 32) return;                                          // (timing taps 32 / 64 / 128: the kernel up to here)
    // ---- B. one leader per lane: the walk
    const int gLow = group_letters(kLow);
    auto startsOf = [](uint32_t M, uint32_t V, uint32_t B) -> uint32_t { return (B & M) | (V & M & (0u - M)); };
    uint32_t need[2] = {0u, 0u}, needK[2] = {0u, 0u};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t L = (uint32_t)t + (uint32_t)r * GTHREADS;
        if (L >= nLead) continue;
        const uint32_t j = sA[L];
        const int dd = (int)sDF[L];
        const unsigned long long hq = getHits(L);
        uint32_t n = 0;
        sSeg[L] = make_uint4(0u, 0u, 0u, 0u);                         // (unused segment words of a record are zero)
        unsigned long long hist = 0, histEnd = 0;                     // segments per LAST level; ends of segments that begin above kLow
        bool split = false, bad = false;
        auto metaAt = [&](uint32_t i) -> uint32_t { const uint32_t x = i - spanLo; bad |= x >= spanN; return (uint32_t)sMeta[x < spanN ? x : spanN]; };
        auto one = [&](uint32_t i, int kLast, uint32_t m) {
            const int dup = (int)(m >> DS);
            const int kFirst = dup < RANGE_LETTERS ? kLow : (dup + 1 > kLow ? dup + 1 : kLow);
            if (kFirst > kLast) return;
            const uint32_t x = i - spanLo;
            const uint32_t s = sTax[x < spanN ? x : 0u] | ((uint32_t)kFirst << 22) | ((uint32_t)kLast << 27);
            if (n < (uint32_t)INL) reinterpret_cast<uint32_t *>(&sSeg[L])[n] = s;
            else {                                                     // a list: the fifth and further segments wait for the pool block
                const uint32_t at = atomicAdd(&sOvfN, 1u);           // (the fourth moves there from the record, by its leader)
                if (at < (uint32_t)OVF) sOvf[at] = make_uint2(s, L | ((n - (uint32_t)(INL - 1)) << 10));
            }
            hist += 1ull << (8 * (kHigh - kLast));
            if (kFirst > kLow) { split = true; histEnd += 1ull << (8 * (kHigh - kFirst + 1)); }
            ++n;
        };
        {   // the merged walk of walk_segments over the staged span
            const uint32_t mj = metaAt(j);
            one(j, dd, mj);
            const int chain0 = dd > RANGE_LETTERS ? dd : RANGE_LETTERS;
            uint32_t li = j, ri = j + 1u;
            uint32_t mri = ri < nIdx ? metaAt(ri) : 0u;
            int lc = li > 0 ? ((int)(mj & LM) < chain0 ? (int)(mj & LM) : chain0) : -1;
            int rc = ri < nIdx ? ((int)(mri & LM) < chain0 ? (int)(mri & LM) : chain0) : -1;
            uint32_t steps = 0;
            while (lc >= gLow || rc >= gLow) {
                if (++steps > G2_STEPS) { bad = true; break; }
                if (lc >= rc) {
                    --li;
                    const uint32_t mli = metaAt(li);
                    one(li, lc < dd ? lc : dd, mli);
                    const int l = (int)(mli & LM);
                    lc = li > 0 ? (l < lc ? l : lc) : -1;
                } else {
                    one(ri, rc < dd ? rc : dd, mri);
                    ++ri;
                    if (ri < nIdx) { mri = metaAt(ri); const int l = (int)(mri & LM); rc = l < rc ? l : rc; } else rc = -1;
                }
            }
        }
        if (bad) atomicOr(&sBail, n > G2_STEPS / 2 ? 1u : 2u);          // (1: a long list; 2: a walk that left the staged span)
        // |T_k| = segments whose last level is k or deeper, less those that begin above k: running sums over the byte fields
        unsigned long long cnt8 = hist * 0x0101010101010101ull;
        if (split) cnt8 -= histEnd * 0x0101010101010101ull;
        uint32_t nlev = 0, V = 0, B = 0, pn = 0, pg = 0;
#pragma unroll
        for (int lv = 0; lv < NL; ++lv) {
            const uint32_t c = (uint32_t)(cnt8 >> (8 * lv)) & 255u, h = (uint32_t)(hq >> (8 * lv)) & 255u;
            nlev |= (c < 7u ? c : 7u) << (3 * lv);
            if (h) { V |= 1u << lv; if (!(pg == h && pn == c)) B |= 1u << lv; }   // a run of levels with the same (|T|, hits): one key
            pn = c; pg = h;                                           // (pg = 0 after a level without hits: the next one begins a run)
        }
        const bool sat = (nlev & (nlev >> 1) & (nlev >> 2) & 0x249249u) != 0u;
        sA[L] = (n & 255u) | (nlev << 8);
        sDF[L] = (uint32_t)dd | (split ? 0x100u : 0u) | (sat ? 0x200u : 0u) | (V << 16) | (B << 24);
        putSizes(L, cnt8);
        if (n > (uint32_t)INL) need[r] = n - (uint32_t)(INL - 1) + 1u + (sat ? POOL_SIZES : 0u);
        if (V) {                                                      // its own keys: the segments of the record (a list's further ones: phase C)
            const uint4 sg = sSeg[L];
            const uint32_t inl = n <= (uint32_t)INL ? n : (uint32_t)INL;   // (a list's fourth segment is still in the record's place)
            uint32_t k = 0;
            if (inl > 0u) k += (uint32_t)__popc(startsOf(seg_level_mask(sg.x, kHigh), V, B));
            if (inl > 1u) k += (uint32_t)__popc(startsOf(seg_level_mask(sg.y, kHigh), V, B));
            if (inl > 2u) k += (uint32_t)__popc(startsOf(seg_level_mask(sg.z, kHigh), V, B));
            if (inl > 3u) k += (uint32_t)__popc(startsOf(seg_level_mask(sg.w, kHigh), V, B));
            needK[r] = k;
        }
    }
    __syncthreads();
    if (flags & 64) return;
    const uint32_t nOvf = sOvfN;
    if (sBail != 0u || nOvf > (uint32_t)OVF) {                    // (uniform) nothing has left the workgroup yet: the tile is group_kernel's
        if (t == 0) slowList[atomicAdd(slowCount, 1u)] = tile | (sBail << 28) | (nOvf > (uint32_t)OVF ? 4u << 28 : 0u);   // (why: diagnostics)
        return;
    }
    // ---- C. one allocation: pool blocks and keys
    uint32_t nkPark = 0;
    for (uint32_t x = t; x < nOvf; x += GTHREADS) {
        const uint2 e = sOvf[x];
        const uint32_t o = e.y & 1023u;
        const uint32_t df = sDF[o];
        nkPark += (uint32_t)__popc(startsOf(seg_level_mask(e.x, kHigh), (df >> 16) & 255u, df >> 24));
    }
    uint32_t offP, offK, totP, totK;
    block_excl_prefix_sum2(need[0] + need[1], needK[0] + needK[1] + nkPark, shU, offP, offK, totP, totK);
    if (t == 0) {
        sBase = NOPOS;
        if constexpr ((VAR & 4) != 0) { if (totP) sBase = 1u + tile * 96u; }   // (timing variant: no cursor -- what do the two same-address atomics of 1.3 M workgroups cost?)
        else
        if (totP) { const unsigned long long at = atomicAdd(poolCursor, (unsigned long long)totP); if (at + totP <= (unsigned long long)poolCap) sBase = (uint32_t)at; }
    }
    if (t == 64) {
        sBaseK = NOPOS;
        if constexpr ((VAR & 4) != 0) { if (totK) sBaseK = (tile * 700u) % (keyCap - 8192u); }
        else
        if (totK) { const unsigned long long at = atomicAdd(keyCursor, (unsigned long long)totK); if (at + totK <= (unsigned long long)keyCap) sBaseK = (uint32_t)at; }
    }
    // a list's place inside the workgroup's pool block is known before the block's own place is: it goes into the leader's
    // words now, so that ONE barrier serves both (the block's base is added where the word is used)
    uint32_t sFourth[2] = {0u, 0u}, relP[2] = {0u, 0u};               // (a list's fourth segment: its keys are its leader's)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t L = (uint32_t)t + (uint32_t)r * GTHREADS;
        if (L >= nLead || need[r] == 0u) continue;
        sFourth[r] = sSeg[L].w;
        relP[r] = offP;
        reinterpret_cast<uint32_t *>(&sSeg[L])[INL - 1] = offP;
        offP += need[r];
    }
    __syncthreads();
    const bool fits = sBase != NOPOS;
    const uint32_t poolBase = fits ? sBase : 0u;
    if (fits) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t L = (uint32_t)t + (uint32_t)r * GTHREADS;
            if (L >= nLead || need[r] == 0u) continue;
            const uint32_t at = poolBase + relP[r];
            const bool sat = (sDF[L] & 0x200u) != 0u;
            pool[at] = sA[L] & 255u;
            if (sat) {
                const unsigned long long c8 = getSizes(L);
#pragma unroll
                for (int q = 0; q < 4; ++q) pool[at + 1u + (uint32_t)q] = ((uint32_t)(c8 >> (16 * q)) & 255u) | (((uint32_t)(c8 >> (16 * q + 8)) & 255u) << 16);
            }
            pool[at + 1u + (sat ? POOL_SIZES : 0u)] = sFourth[r];    // the list's first segment of the pool block
        }
    }
    if constexpr ((VAR & 2) != 0) __syncthreads();
    if (flags & 256) return;                                         // (timing tap: everything up to the allocation)
    // ---- D. sorted layout: the records
    if (cellW == 16u) {
        // WHOLE 64-byte cells: a lane owns a record, FOUR lanes store its cell in one instruction -- the record's two halves and 32
        // bytes of zeros -- so that a store instruction touches 16 whole cells instead of halves of 64: a random partial write of
        // a cell costs the memory side about three whole ones (tools/scatter_probe.hip: 27 against 77 G records/s).  The first half
        // passes through the dead index span (a wavefront's 64 at a time), the second is the leader's words where they lie.
        uint4 *sStage = reinterpret_cast<uint4 *>(sRaw) + wv * 64;
        uint4 *rec4 = reinterpret_cast<uint4 *>(rec);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint4 o0 = make_uint4(base + (uint32_t)i, fmax[i], w2[i], 0u);
            uint32_t lw = NOPOS;                                      // the query's leader | "a list" << 31
            if (d[i] != 0) {
                const uint32_t df = sDF[myL[i]];
                o0.w = sA[myL[i]];
                o0.z |= ((df & 0x100u) ? REC_SPLIT : 0u) | ((df & 0x200u) ? REC_SAT : 0u);
                lw = myL[i] | (((o0.w & 255u) > (uint32_t)INL) ? 0x80000000u : 0u);
            }
            sStage[lane] = o0;
            LDS_WAVE_SYNC_G();
#pragma unroll
            for (int k = 0; k < 4; ++k) {                            // instruction k: the cells of lanes 16k .. 16k+15, a quad each
                const int src = 16 * k + (lane >> 2);
                const uint32_t sk = (uint32_t)__shfl((int)slot[i], src), lk = (uint32_t)__shfl((int)lw, src);
                const uint32_t part = (uint32_t)lane & 3u;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (part == 0u) v = sStage[src];
                else if (part == 1u && lk != NOPOS) { v = sSeg[lk & 1023u]; if (lk >> 31) v.w += poolBase; }
                if (sk != NOPOS) rec4[(size_t)sk * 4u + part] = v;
            }
            LDS_WAVE_SYNC_G();
        }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (slot[i] == NOPOS) continue;
        uint4 o0 = make_uint4(base + (uint32_t)i, fmax[i], w2[i], 0u), o1 = make_uint4(0u, 0u, 0u, 0u);
        if (d[i] != 0) {
            const uint32_t df = sDF[myL[i]];
            o0.w = sA[myL[i]];
            o0.z |= ((df & 0x100u) ? REC_SPLIT : 0u) | ((df & 0x200u) ? REC_SAT : 0u);
            o1 = sSeg[myL[i]];
            if ((o0.w & 255u) > (uint32_t)INL) o1.w += poolBase;     // (a list: [3] = its pool block)
        }
        uint4 *o = reinterpret_cast<uint4 *>(rec + (size_t)slot[i] * 8u);
        o[0] = o0; o[1] = o1;
    }
    }
    if (fits)
        for (uint32_t x = t; x < nOvf; x += GTHREADS) {
            const uint2 e = sOvf[x];
            const uint32_t o = e.y & 1023u;
            pool[poolBase + sSeg[o].w + 1u + ((sDF[o] & 0x200u) ? POOL_SIZES : 0u) + (e.y >> 10)] = e.x;
        }
    // ---- E. the profile keys (as group_kernel: one key {levels, |T|, taxon, hits} per run of levels of a segment)
    if (totK == 0u || sBaseK == NOPOS || (flags & 128)) return;      // (uniform)
    if (cellW == 16u) __syncthreads();                               // (the records' first halves have left the span)
    constexpr uint32_t KSTAGE = (uint32_t)(SPAN_BYTES / 8);
    const bool staged = totK <= KSTAGE;
    unsigned long long *sKeys = reinterpret_cast<unsigned long long *>(sRaw);   // the index span is dead
    uint32_t kw = offK + (staged ? 0u : sBaseK);
    auto cutRuns = [&](uint32_t sg, uint32_t V, uint32_t B, unsigned long long hq, unsigned long long c8) {
        const uint32_t M = seg_level_mask(sg, kHigh);
        uint32_t starts = (B & M) | (V & M & (0u - M));
        const uint32_t stops = (B | ~V) & M;                         // a run ends before the next run's start, a level without hits, or with the segment
        while (starts) {
            const int lo = __ffs((int)starts) - 1;
            starts &= starts - 1u;
            const uint32_t beyond = stops & ~((2u << lo) - 1u);
            const int hi = beyond ? __ffs((int)beyond) - 2 : 31 - __clz((int)M);
            const uint32_t n = (uint32_t)(c8 >> (8 * lo)) & 255u, hits = (uint32_t)(hq >> (8 * lo)) & 255u, tx = sg & SEG_TAX_MASK;
            if (flags & 16) { ++kw; continue; }
            // (|T| <= G2_STEPS + 1 and < nTaxa: it always fits the key's field of min(tb, 13) bits -- no straight-to-the-tables path here)
            const unsigned long long key = group_key((uint32_t)lo, (uint32_t)hi, n, tx, hits);
            if (staged) sKeys[kw++] = key; else profKeys[kw++] = key;
        }
    };
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t L = (uint32_t)t + (uint32_t)r * GTHREADS;
        if (L >= nLead || needK[r] == 0u) continue;
        const uint32_t df = sDF[L], V = (df >> 16) & 255u, B = df >> 24;
        const unsigned long long hq = getHits(L), c8 = getSizes(L);
        const uint4 sg = sSeg[L];
        const uint32_t n = sA[L] & 255u, w4 = n > (uint32_t)INL ? sFourth[r] : sg.w, nn = n < (uint32_t)INL ? n : (uint32_t)INL;
        if constexpr ((VAR & 1) != 0) {
            if (nn > 0u) cutRuns(sg.x, V, B, hq, c8);
            if (nn > 1u) cutRuns(sg.y, V, B, hq, c8);
            if (nn > 2u) cutRuns(sg.z, V, B, hq, c8);
            if (nn > 3u) cutRuns(w4, V, B, hq, c8);
        } else {
#pragma unroll 1
            for (uint32_t q = 0; q < nn; ++q) cutRuns(q == 0u ? sg.x : q == 1u ? sg.y : q == 2u ? sg.z : w4, V, B, hq, c8);   // (one copy of the loop body: code size)
        }
    }
    for (uint32_t x = t; x < nOvf; x += GTHREADS) {
        const uint2 e = sOvf[x];
        const uint32_t o = e.y & 1023u;
        const uint32_t df = sDF[o];
        if ((df >> 16) & 255u) cutRuns(e.x, (df >> 16) & 255u, df >> 24, getHits(o), getSizes(o));
    }
    if (staged && !(flags & 16)) {                                   // (uniform)
        __syncthreads();
        for (uint32_t x = t; x < totK; x += GTHREADS) profKeys[(size_t)sBaseK + x] = sKeys[x];
    }
}

static constexpr int FTA = 2;       // taxa kept in registers
static constexpr int RMAX = 1024;   // longest staging row row_merge_kernel handles; the fast kernels hand reads with longer rows to score_kernel
static constexpr uint32_t ROW_MERGE = 0x80000000u;   // rowLen flag: the row holds records, not final {taxon, score} pairs

// One wavefront working alone on LDS: LDS instructions of a wave execute in order, so only the compiler has
// to be kept from moving them across the point (a workgroup barrier would also drain pending global stores).
#define LDS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// A staging record (8 bytes).  x = taxon (20 bits) | ... | kind << 30
//   kind 0  event:        x |= level << 23; y = |T| << 16 | hits.  One (event, taxon) contribution, in the read's flush order.
//   kind 1  final score:  y = float bits (a register taxon)
//   kind 2  profile only: as kind 0, but the score is already in a register taxon's final score
//   kind 3  segment:      x |= kFirst << 20 | RK_SEG_DESC; y = kLast | |T_kFirst| << 5 | |T_kLast| << 18.  All events of one
//                         taxon segment of one query: its one or two levels kFirst..kLast in ascending order of k
//                         (descending with RK_SEG_DESC), one hit each; the sizes are 13 bits each
static constexpr uint32_t RK_FINAL = 1u << 30, RK_PROFILE = 2u << 30, RK_SEG = 3u << 30, RK_SEG_DESC = 1u << 29;
__device__ __forceinline__ uint32_t seg_size(uint32_t y, bool first) { return (first ? (y >> 5) : (y >> 18)) & 0x1FFFu; }
static constexpr int RK_LV_SHIFT = 23;
static constexpr uint32_t RK_LV_MASK = 31u;
__device__ __forceinline__ uint32_t rk_level(uint32_t x) { return (x >> RK_LV_SHIFT) & RK_LV_MASK; }
__device__ __forceinline__ uint64_t profile_key(uint2 e, ProfLayout L)
{
    const uint64_t f = ((uint64_t)rk_level(e.x) << (L.tb + L.nb)) | ((uint64_t)(e.y >> 16) << L.tb) | (e.x & 0xFFFFFu);
    return (f << 16) | (e.y & 0xFFFFu);
}

__device__ __forceinline__ float event_score(int k, uint32_t n)
{
    const float w = (float)(k * k) / 625.0f;                       // Compare.hpp:392
    return __fmul_rn(w, __fdiv_rn(1.0f, (float)n));                // Compare.hpp:924
}

// profile keys a staging record yields
__device__ __forceinline__ uint32_t record_keys(uint2 e)
{
    const uint32_t kind = e.x >> 30;
    return kind == 1u ? 0u : (kind == 3u ? (e.y & 31u) - ((e.x >> 20) & 31u) + 1u : 1u);
}

// The same two correctly rounded divisions, done once per workgroup: w_k for every k and 1/n for n < EV_INV.
static constexpr int EV_INV = 64;
struct EventTables { float w[32]; float inv[EV_INV]; };
__device__ __forceinline__ void event_tables_init(EventTables &T)
{
    for (int i = threadIdx.x; i < 32; i += blockDim.x) T.w[i] = (float)(i * i) / 625.0f;
    for (int i = threadIdx.x; i < EV_INV; i += blockDim.x) T.inv[i] = i ? __fdiv_rn(1.0f, (float)i) : 0.0f;
    __syncthreads();
}
__device__ __forceinline__ float event_score(const EventTables &T, int k, uint32_t n)
{
    const float inv = (n < (uint32_t)EV_INV) ? T.inv[n] : __fdiv_rn(1.0f, (float)n);
    return __fmul_rn(T.w[k & 31], inv);
}

// ------------------------------------------------------------------------------------------------
// score
// ------------------------------------------------------------------------------------------------
struct ScoreArgs {
    const uint32_t *rec; const uint64_t *kmerOff; const uint32_t *pool;   // records by slot; slots of read r: kmerOff[r] .. kmerOff[r+1]
    uint32_t rowPerQuery;                        // fast kernels: a read's staging row may hold max(RMAX, rowPerQuery * its k-mers) records (0: RMAX) -- long reads have long rows
    uint32_t recQS;                              // log2 of the 16-byte words from one record to the next (a shift, not a multiply, in the hot loops)
    uint32_t recCW;                              // words from one record to the next (kasa_ctx::recCW: RW, or 16 for narrow records in 64-byte cells)
    uint32_t nReads; int kHigh, kLow; uint32_t nTaxa;
    float *scratch;                              // per block: nTaxa floats, all zero between reads
    uint64_t *cntUnique, *cntAllHi, *cntAllMid, *cntAllLo;
    uint32_t *rowPos, *rowLen; uint2 *st; uint32_t stCap; unsigned long long *stCursor;   // staging rows: {taxon, score bits}
    uint32_t *errFlag; int wantPerRead;
    int addProfile;                              // 0 on a rerun that only re-emits rows
    const uint32_t *list; uint32_t nList;        // general kernel: reads to process (NULL = all)
    const uint32_t *flushPos; const uint64_t *flushOff;   // general kernel: F per level of the listed reads' queries; first query of list entry wi
    uint32_t *fbList, *fbCount;                  // fast kernel: reads it hands to the general kernel
    uint32_t *ovList, *ovCount;                  // general kernel, first pass: reads it hands to the second pass (NULL = last pass)
    uint32_t *mainOut;                           // fast kernels: {taxon 0, taxon 1, position, count} per read, main -> other
    uint32_t *otherOff64;                        // fast kernels: records of the other taxa a read has yielded before slot 64 i
    uint32_t *rowKey; uint32_t keyCap; unsigned long long *keyCursor;   // fast kernels: profile-key slots of a read's row (key buffer, its capacity, its cursor)
    uint32_t nQ;
    uint32_t *why;                               // fast kernel: fallback reasons (diagnostics)
    uint32_t *workCursor;                        // fast kernel: next read a wavefront takes
    int forceHandOn;                             // general kernel, first pass: every read goes to the second pass (test tap 16384)
    uint32_t *gwin; uint32_t gwinCap;            // general kernel, third pass (GWIN): pending windows in device memory, per block 4 arrays of gwinCap words
};

static constexpr int AGG = 256;                                   // per-read profile aggregation table (LDS)
static constexpr int PCAP_SMALL = 96;                              // pending window of the first pass (small LDS footprint: many wavefronts per CU)
static constexpr unsigned long long AGG_EMPTY = ~0ull;

__device__ __forceinline__ void profile_add(const ScoreArgs &A, int lv, uint32_t tx, uint32_t n, unsigned long long c)
{
    const size_t cell = (size_t)lv * A.nTaxa + tx;
    if (n == 1) atomicAdd((unsigned long long *)&A.cntUnique[cell], c);
    fixed_add(A.cntAllHi, A.cntAllMid, A.cntAllLo, cell, c, n);
}

// Flush positions of single queries, for the reads the general kernel replays: F_k(p) for every level of every query of
// the listed reads, recomputed the way group_kernel computes them (scan of the rest of p's tile, then tileNext).  The
// records keep only the ORDER of a query's own events; the general kernel also needs the order between queries.
// One wavefront per query; out[(flushOff[wi] + j) * nK + lv].
template <int RW, class Key>
__global__ __launch_bounds__(256) void flush_positions_kernel(
    const uint32_t *__restrict__ list, uint32_t nList, const uint64_t *__restrict__ flushOff, const uint64_t *__restrict__ kmerOff,
    const uint32_t *__restrict__ rec, const Key *__restrict__ qKmer, const uint8_t *__restrict__ depth, uint32_t nQ,
    const uint32_t *__restrict__ tileNext, uint32_t nTiles, int kHigh, int kLow, uint32_t *__restrict__ out, uint32_t recCW)
{
    const int nK = kHigh - kLow + 1;
    const uint32_t allLv = (nK >= 32) ? 0xFFFFFFFFu : ((1u << nK) - 1u);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t wi = blockIdx.x; wi < nList; wi += gridDim.x) {
        const uint32_t r = list ? list[wi] : wi;
        const uint64_t o0 = kmerOff[r];
        const uint32_t cnt = (uint32_t)(kmerOff[r + 1] - o0);
        const uint64_t f0 = flushOff[wi];
        for (uint32_t j = wv; j < cnt; j += 4) {
            const uint32_t *w = rec + (o0 + j) * recCW;
            const uint32_t p = w[0];
            const int d = (int)(w[2] & 31u);
            uint32_t myF = NOPOS;                                     // lane lv holds F of level lv
            if (d > 0) {
                const uint32_t tile = p / TILE;
                const uint32_t tileEnd = ((uint64_t)(tile + 1) * TILE < nQ) ? (tile + 1) * TILE : nQ;
                uint32_t todo = allLv & ~((1u << (kHigh - d)) - 1u);  // levels kLow..d
                for (uint32_t b0 = p + 1; b0 < tileEnd && todo; b0 += 64) {
                    const uint32_t pp = b0 + lane;
                    uint32_t m = 0;
                    if (pp < tileEnd) {
                        const int ql = lcp_letters<Key>(qKmer[pp - 1], qKmer[pp]);
                        m = special_mask(ql, (int)depth[pp], kHigh, allLv);
                    }
                    for (int lv = 0; lv < nK; ++lv) {
                        if (!((todo >> lv) & 1u)) continue;
                        const unsigned long long b = __ballot((m >> lv) & 1u);
                        if (b) { if (lane == lv) myF = b0 + (uint32_t)(__ffsll((long long)b) - 1); todo &= ~(1u << lv); }
                    }
                }
                if (lane < nK && ((todo >> lane) & 1u)) myF = tileNext[(size_t)lane * nTiles + tile];
            }
            if (lane < nK) out[(f0 + j) * nK + lane] = myF;
        }
    }
}

// General kernel: one wavefront per read, events replayed in flush order through a pending window -- any number of
// levels, any taxon-set size, groups that stay open across the read's later queries (equal (F, k) = same group: its
// hit count grows).  PC = capacity of the pending window.  The kernel is latency-bound (dependent gathers per query and
// per taxon list), so the first pass runs with a small window -- little LDS, many resident wavefronts -- and hands the
// rare read that overflows it (or the aggregation table) to a second pass with the full window.  A read that is handed
// on leaves nothing behind: its score cells are cleared again and its profile counts never left LDS.
// DENSE: the read's score row lies in LDS (dynamic: nTaxa floats) and an event's taxa are dealt out to the LANES -- a query of
// a conserved region brings hundreds of taxa per level, and the lane-owns-its-cells form below walks all of them on every
// lane for every event.  The segments of the query being replayed wait in LDS with the sizes of its levels (a query's
// events mostly follow each other).  Within an event every taxon occurs once (a taxon's segments cover disjoint levels), so
// the lanes' read-modify-writes never meet; LDS operations of a wavefront execute in order, so events do not overtake.
static constexpr int DENSE_TAXA = 16384;      // 64 KB of LDS for the row at most
static constexpr int DQ_SEGS = 1024;          // segments of one query kept in LDS
// GWIN (third pass, narrow records): the pending window lives in device memory instead of LDS -- as long as the read has
// queries times levels, so it cannot overflow.  Every access is a round trip past the L1 (agent-scope atomics, a fence where
// lanes read what other lanes wrote): slow, exact, and only for reads that keep more than PCAP groups pending.
template <int PC, int RW, bool DENSE = false, bool GWIN = false>
__global__ __launch_bounds__(64) void score_kernel(ScoreArgs A)
{
    typedef RecTraits<RW> RT;
    extern __shared__ float sRowDyn[];
    __shared__ uint32_t sSegQ[DENSE ? DQ_SEGS : 1];
    __shared__ uint32_t sCntQ[DENSE ? 32 : 1];
    __shared__ uint32_t sPreQ[DENSE ? 32 : 1];     // segments of the staged query whose last level reaches level lv: they are the FIRST ones of its list
    // (the dense form of narrow records keeps no profile -- group_stage's -- and no list of touched taxa: the LDS they would
    // take is what limits the resident wavefronts of this latency-bound kernel)
    constexpr bool LEAN = DENSE && GpOf<RW>::v;
    // (64-byte records, first pass: 19 levels make more distinct (level, |T|, taxon) counts per read than 256 -- at C3 4749 of the
    // 5663 reads of this kernel overflowed the table and took the second pass, 12 ms at two wavefronts per CU)
    constexpr int AGGN = LEAN ? 1 : (RW == 16 && PC == PCAP_SMALL) ? 4 * AGG : AGG, TLN = LEAN ? 1 : TLIST;
    constexpr int AGGB = AGGN == 4 * AGG ? 10 : 8;
    static_assert(AGG == 256, "aggAdd's hash takes log2(AGGN) bits");
    __shared__ unsigned long long aKey[AGGN];
    __shared__ uint32_t aCnt[AGGN];
    __shared__ uint32_t pF[PC], pRef[PC];
    __shared__ uint32_t pCnt[PC];
    __shared__ uint8_t pK[PC];
    // the pending window: entry e = {flush position, slot of its query, hits, level}; in LDS, or (GWIN) in this block's piece of A.gwin
    uint32_t *gw = GWIN ? A.gwin + (size_t)blockIdx.x * 4u * A.gwinCap : nullptr;
    const int pcap = GWIN ? (int)A.gwinCap : PC;
    auto wGet = [&](int field, int e) -> uint32_t {
        if constexpr (GWIN) return __hip_atomic_load(gw + (size_t)field * A.gwinCap + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return field == 0 ? pF[e] : field == 1 ? pRef[e] : field == 2 ? pCnt[e] : (uint32_t)pK[e];
    };
    auto wPut = [&](int field, int e, uint32_t v) {
        if constexpr (GWIN) __hip_atomic_store(gw + (size_t)field * A.gwinCap + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { if (field == 0) pF[e] = v; else if (field == 1) pRef[e] = v; else if (field == 2) pCnt[e] = v; else pK[e] = (uint8_t)v; }
    };
    auto wSync = [&]() { if constexpr (GWIN) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __syncthreads(); };   // lanes read what other lanes wrote
    __shared__ uint32_t sTouched;
    __shared__ uint32_t sList[TLN];
    const int lane = threadIdx.x;
    const int nK = A.kHigh - A.kLow + 1;
    float *score = DENSE ? sRowDyn : A.scratch + (size_t)blockIdx.x * A.nTaxa;
    if constexpr (DENSE) for (uint32_t tx = lane; tx < A.nTaxa; tx += 64) score[tx] = 0.0f;
    uint32_t cachedSlot = NOPOS, cachedN = 0;     // DENSE: the query whose segments and level sizes lie in LDS (uniform)
    uint32_t mySeg = 0;                            // DENSE, a query of at most 64 segments: lane i holds segment i (no staging in LDS)
    __shared__ EventTables evT;
    event_tables_init(evT);

    for (int i = lane; i < AGGN; i += 64) { aKey[i] = AGG_EMPTY; aCnt[i] = 0u; }
    const uint32_t nWork = A.list ? A.nList : A.nReads;
    for (uint32_t wi = blockIdx.x; wi < nWork; wi += gridDim.x) {
        const uint32_t r = A.list ? A.list[wi] : wi;
        const uint64_t o0 = A.kmerOff[r];
        const uint32_t cnt = (uint32_t)(A.kmerOff[r + 1] - o0);
        const uint64_t f0 = A.flushOff[wi];
        int head = 0, tail = 0;            // pending window [head, tail), sorted by (F, k) ascending
        uint32_t cTax = 0xFFFFFFFFu;       // this lane's cached score cell (taxa with tx % 64 == lane live here)
        float cVal = 0.0f;
        const bool mayHandOn = A.ovList != nullptr;
        bool ovf = mayHandOn && A.forceHandOn != 0;   // this read does not fit this pass (per lane; combined with a ballot)
        if (lane == 0) sTouched = 0;
        __syncthreads();

        // c hits for the score cell of taxon tx; called by the lane that owns the cell (tx % 64 == lane).  The lane keeps
        // the cell it touched last in a register: a read's events go almost all to one or two taxa, and a
        // load-add-store chain through global memory per event would cost a round trip each.
        auto cellAdd = [&](const uint32_t tx, const float sc, const uint32_t c) {
            if (cTax != tx) {
                if (cTax != 0xFFFFFFFFu) score[cTax] = cVal;
                cVal = score[tx];
                cTax = tx;
                if (cVal == 0.0f) { const uint32_t ti = atomicAdd(&sTouched, 1u); if (ti < (uint32_t)TLIST) sList[ti] = tx; }
            }
            for (uint32_t j = 0; j < c; ++j) cVal = __fadd_rn(cVal, sc);      // Compare.hpp:528-530, one add per hit
        };
        // profile counts of this read are summed per (level, |T|, taxon) in LDS first and leave as one set of atomics
        // per distinct key at the end of the read (device-scope atomics are the scarce resource); any lane may call it
        auto aggAdd = [&](const int lv, const uint32_t n, const uint32_t tx, const uint32_t c) {
            const unsigned long long key = ((unsigned long long)lv << 56) | ((unsigned long long)n << 32) | tx;
            uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - AGGB)) & (uint32_t)(AGGN - 1);
            for (int probe = 0; probe < 16; ++probe, h = (h + 1) & (uint32_t)(AGGN - 1)) {
                const unsigned long long seen = atomicCAS(&aKey[h], AGG_EMPTY, key);
                if (seen == AGG_EMPTY || seen == key) { atomicAdd(&aCnt[h], c); return; }
            }
            if (mayHandOn) ovf = true; else profile_add(A, lv, tx, n, c);
        };
        // one flushed group: level k of the query at `slot`, c hits of this read; all lanes call it with the same values.
        // |T| = the segments covering k; every taxon is handled by the lane that owns its cell.
        auto applyVals = [&](const int k, const uint32_t slot, const uint32_t c) {
            const uint32_t *w = A.rec + (size_t)slot * A.recCW;
            if constexpr (DENSE) {
                // A query of at most 64 segments (nearly all of them: two or three as a rule) is a register per lane: |T_k| is a
                // ballot, its score a table look-up -- the staging below (LDS marks, two running sums, four wave syncs) and a float
                // division per event were 700 instructions per query, and a read of this kernel is ONE wavefront's (the long-read
                // workload: 12 % of 10 kb reads repeat a prefix of their own; round 6: 196 -> see DESIGN 3d)
                if (slot != cachedSlot && rec_nseg<RW>(w, A.pool) <= 64u) {
                    cachedN = rec_nseg<RW>(w, A.pool);
                    cachedSlot = slot;
                    mySeg = (uint32_t)lane < cachedN ? rec_seg<RW>(w, A.pool, cachedN, (uint32_t)lane) : 0u;
                }
                if (slot == cachedSlot && cachedN <= 64u) {
                    const bool cov = (uint32_t)lane < cachedN && seg_covers(mySeg, (uint32_t)k);
                    const uint32_t n = (uint32_t)__popcll(__ballot(cov));
                    const float sc = event_score(evT, k, n);
                    if (cov) {
                        const uint32_t tx = mySeg & SEG_TAX_MASK;
                        if (A.wantPerRead) { float v = score[tx]; for (uint32_t j = 0; j < c; ++j) v = __fadd_rn(v, sc); score[tx] = v; }
                        if (A.addProfile) aggAdd(A.kHigh - k, n, tx, c);
                    }
                    LDS_WAVE_SYNC();
                    return;
                }
                if (slot != cachedSlot) {                                    // stage the query: segments, +1 / -1 at the ends of their level ranges, running sum
                    const uint32_t ns = rec_nseg<RW>(w, A.pool);
                    cachedN = ns;
                    cachedSlot = slot;
                    if (lane < 32) { sCntQ[lane] = 0u; sPreQ[lane] = 0u; }
                    LDS_WAVE_SYNC();
                    for (uint32_t b0 = 0; b0 < ns; b0 += 256) {                  // four loads in flight per lane: the kernel waits for memory, not for arithmetic
                        uint32_t sg4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const uint32_t i = b0 + 64u * u + lane; sg4[u] = i < ns ? rec_seg<RW>(w, A.pool, ns, i) : 0u; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t i = b0 + 64u * u + lane;
                            if (i < ns) {
                                const uint32_t sg = sg4[u];
                                if (i < (uint32_t)DQ_SEGS) sSegQ[i] = sg;
                                atomicAdd(&sCntQ[A.kHigh - (int)(sg >> 27)], 1u);
                                atomicSub(&sCntQ[A.kHigh - (int)((sg >> 22) & 31u) + 1], 1u);
                                atomicAdd(&sPreQ[A.kHigh - (int)(sg >> 27)], 1u);
                            }
                        }
                    }
                    LDS_WAVE_SYNC();
                    uint32_t v = lane < 32 ? sCntQ[lane] : 0u, pv = lane < 32 ? sPreQ[lane] : 0u;
                    v = wave_incl_sum(v);
                    pv = wave_incl_sum(pv);
                    LDS_WAVE_SYNC();
                    if (lane < 32) { sCntQ[lane] = v; sPreQ[lane] = pv; }
                    LDS_WAVE_SYNC();
                }
                const int lv = A.kHigh - k;
                // the list comes in descending order of the segments' last level: those that reach level k are its first
                // sPreQ[lv] -- an event looks at no more (a conserved k-mer's list: hundreds of shallow chance matches behind a
                // handful of deep ones)
                const uint32_t n = sCntQ[lv], ns = sPreQ[lv];
                const float sc = event_score(k, n);
                for (uint32_t b0 = 0; b0 < ns; b0 += 64) {
                    const uint32_t i = b0 + lane;
                    if (i >= ns) continue;
                    const uint32_t sg = i < (uint32_t)DQ_SEGS ? sSegQ[i] : rec_seg<RW>(w, A.pool, cachedN, i);
                    if (!seg_covers(sg, (uint32_t)k)) continue;
                    const uint32_t tx = sg & SEG_TAX_MASK;
                    if (A.wantPerRead) { float v = score[tx]; for (uint32_t j = 0; j < c; ++j) v = __fadd_rn(v, sc); score[tx] = v; }
                    if (A.addProfile) aggAdd(lv, n, tx, c);
                }
                LDS_WAVE_SYNC();
                return;
            }
            const uint32_t nseg = rec_nseg<RW>(w, A.pool);
            uint32_t n = 0;
            for (uint32_t b0 = 0; b0 < nseg; b0 += 64) {
                const bool in = b0 + lane < nseg;
                const uint32_t s = in ? rec_seg<RW>(w, A.pool, nseg, b0 + lane) : 0u;
                n += (uint32_t)__popcll(__ballot(in && seg_covers(s, (uint32_t)k)));
            }
            const float sc = event_score(k, n);
            const int lv = A.kHigh - k;
            for (uint32_t i = 0; i < nseg; ++i) {
                const uint32_t s = rec_seg<RW>(w, A.pool, nseg, i);
                if (!seg_covers(s, (uint32_t)k)) continue;
                const uint32_t tx = s & SEG_TAX_MASK;
                if ((tx & 63u) != (uint32_t)lane) continue;                  // a cell always lives on one lane
                if (A.wantPerRead) cellAdd(tx, sc, c);
                if (A.addProfile) aggAdd(lv, n, tx, c);
            }
        };
        auto apply = [&](int e) { applyVals((int)wGet(3, e), wGet(1, e), wGet(2, e)); };   // the pending entry at `e`

        uint32_t pnext = cnt ? A.rec[(size_t)o0 * A.recCW] : 0u;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t p = pnext;
            const uint32_t slot = (uint32_t)(o0 + j);
            pnext = (j + 1 < cnt) ? A.rec[(size_t)(slot + 1) * A.recCW] : 0xFFFFFFFFu;
            if (mayHandOn && __ballot(ovf) != 0ull) { ovf = true; break; }
            // everything that flushes at or before p precedes all events of this and later queries
            while (head < tail && wGet(0, head) <= p) { apply(head); ++head; }
            if (head == tail) head = tail = 0;
            const int d = (int)(A.rec[(size_t)slot * A.recCW + 2] & 31u);
            if (d == 0) continue;
            // lane lv holds the event of level k = kHigh - lv
            const bool has = lane < nK && (A.kHigh - lane) <= d;
            const uint32_t myF = has ? A.flushPos[(f0 + j) * nK + lane] : 0u;
            // The usual case: nothing is pending and every group of this query closes before the read's next query.
            // Then its events need no window: they are replayed at once in (F, k) order.
            if (head == tail && __ballot(has && myF > pnext) == 0ull) {
                uint32_t rank = 0;
                for (int o = 0; o < nK; ++o) {
                    const uint32_t oF = __shfl(myF, o);
                    const bool oHas = (A.kHigh - o) <= d;
                    if (oHas && (oF < myF || (oF == myF && o > lane))) ++rank;
                }
                const int nEv = d - A.kLow + 1;
                for (int rk = 0; rk < nEv; ++rk) {
                    const int src = __ffsll((long long)__ballot(has && rank == (uint32_t)rk)) - 1;
                    applyVals(A.kHigh - src, slot, 1u);
                }
                continue;
            }
            for (int lv = nK - 1; lv >= 0; --lv) {                            // k ascending
                const int k = A.kHigh - lv;
                if (k > d) break;
                const uint32_t F = __shfl(myF, lv);
                // position of (F, k) in the sorted window; equal key = same group: bump its count
                int pos = tail;
                bool same = false;
                for (int b0 = head; b0 < tail; b0 += 64) {
                    const int e = b0 + lane;
                    bool ge = false, eq = false;
                    if (e < tail) {
                        const uint32_t eF = wGet(0, e); const int eK = (int)wGet(3, e);
                        ge = (eF > F) || (eF == F && eK >= k);
                        eq = (eF == F && eK == k);
                    }
                    const unsigned long long m = __ballot(ge);
                    if (m) { pos = b0 + __ffsll((long long)m) - 1; same = (__ballot(eq) != 0ull); break; }
                }
                if (same) {
                    if (lane == 0) wPut(2, pos, wGet(2, pos) + 1u);
                    wSync();
                    continue;
                }
                if (tail >= pcap) {
                    if (head > 0) {                                           // compact the window to the front
                        for (int b0 = head; b0 < tail; b0 += 64) {
                            const int e = b0 + lane;
                            uint32_t f = 0, rf = 0, cc = 0, kk = 0;
                            if (e < tail) { f = wGet(0, e); rf = wGet(1, e); cc = wGet(2, e); kk = wGet(3, e); }
                            wSync();
                            if (e < tail) { wPut(0, e - head, f); wPut(1, e - head, rf); wPut(2, e - head, cc); wPut(3, e - head, kk); }
                            wSync();
                        }
                        pos -= head; tail -= head; head = 0;
                    }
                    if (tail >= pcap) { if (mayHandOn) ovf = true; else if (lane == 0) atomicOr(A.errFlag, 2u); continue; }
                }
                for (int hi = tail; hi > pos; hi -= 64) {                     // shift [pos, tail) right by one
                    const int e = hi - 1 - lane;
                    uint32_t f = 0, rf = 0, cc = 0, kk = 0;
                    if (e >= pos) { f = wGet(0, e); rf = wGet(1, e); cc = wGet(2, e); kk = wGet(3, e); }
                    wSync();
                    if (e >= pos) { wPut(0, e + 1, f); wPut(1, e + 1, rf); wPut(2, e + 1, cc); wPut(3, e + 1, kk); }
                    wSync();
                }
                if (lane == 0) { wPut(0, pos, F); wPut(1, pos, slot); wPut(2, pos, 1u); wPut(3, pos, (uint32_t)k); }
                ++tail;
                wSync();
            }
        }
        ovf = mayHandOn && (__ballot(ovf) != 0ull);
        if (!ovf) while (head < tail) { apply(head); ++head; }
        ovf = mayHandOn && (__ballot(ovf) != 0ull);
        if (ovf) {                                                            // hand the read on, leave no trace
            __syncthreads();
            const uint32_t m = sTouched;
            if (!DENSE && m <= (uint32_t)TLIST) { for (uint32_t i = lane; i < m; i += 64) score[sList[i]] = 0.0f; }
            else { for (uint32_t tx = lane; tx < A.nTaxa; tx += 64) score[tx] = 0.0f; }
            for (int i = lane; i < AGGN; i += 64) { aKey[i] = AGG_EMPTY; aCnt[i] = 0u; }
            if (lane == 0) A.ovList[atomicAdd(A.ovCount, 1u)] = r;
            __syncthreads();
            continue;
        }
        if (cTax != 0xFFFFFFFFu) score[cTax] = cVal;
        __threadfence_block();
        __syncthreads();
        if (A.addProfile)
            for (int i = lane; i < AGGN; i += 64) {
                const unsigned long long key = aKey[i];
                if (key == AGG_EMPTY) continue;
                profile_add(A, (int)(key >> 56), (uint32_t)key, (uint32_t)(key >> 32) & 0xFFFFFFu, aCnt[i]);
                aKey[i] = AGG_EMPTY; aCnt[i] = 0u;
            }

        // ---- emit the row (taxon ascending) and clear the dense row
        if (A.wantPerRead) {
            uint32_t m = sTouched;
            if constexpr (DENSE) {                                           // the row's taxa are counted, not listed
                m = 0;
                for (uint32_t b0 = 0; b0 < A.nTaxa; b0 += 64) { const uint32_t tx = b0 + lane; m += (uint32_t)__popcll(__ballot(tx < A.nTaxa && score[tx] > 0.0f)); }
            }
            unsigned long long start = 0;
            if (lane == 0) {
                start = m ? atomicAdd(A.stCursor, (unsigned long long)m) : 0ull;
                const bool fits = start + m <= (unsigned long long)A.stCap;
                A.rowPos[r] = fits ? (uint32_t)start : 0u; A.rowLen[r] = fits ? m : 0u;
                if (!fits) start = ~0ull;
            }
            start = lane_value<0>(start);
            __threadfence_block();
            if (m && start != ~0ull) {
                if (!DENSE && m <= 64) {
                    const uint32_t mine = (lane < (int)m) ? sList[lane] : 0xFFFFFFFFu;
                    uint32_t rank = 0;
                    for (uint32_t i = 0; i < m; ++i) rank += (__shfl(mine, (int)i) < mine) ? 1u : 0u;
                    if (lane < (int)m) A.st[start + rank] = make_uint2(mine, __float_as_uint(score[mine]));
                } else {
                    uint64_t w = start;
                    for (uint32_t b0 = 0; b0 < A.nTaxa; b0 += 64) {
                        const uint32_t tx = b0 + lane;
                        const float v = (tx < A.nTaxa) ? score[tx] : 0.0f;
                        const unsigned long long mk = __ballot(v > 0.0f);
                        if (v > 0.0f) {
                            const uint64_t o = w + __popcll(mk & ((1ull << lane) - 1ull));
                            A.st[o] = make_uint2(tx, __float_as_uint(v));
                        }
                        w += __popcll(mk);
                    }
                }
            }
            __syncthreads();
            if (!DENSE && m <= (uint32_t)TLIST) { for (uint32_t i = lane; i < m; i += 64) score[sList[i]] = 0.0f; }
            else { for (uint32_t tx = lane; tx < A.nTaxa; tx += 64) score[tx] = 0.0f; }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// score, fast path.  Two kernels over the same records:
//
//   score_main_kernel   one LANE per read.  The replay of a read's events is a sequential float chain per (read,
//       taxon), so 64 reads run side by side in a wavefront.  A read's records lie in its slots in sorted order; the
//       lane streams them front to back.  As long as every query's groups are closed before the read's next matched
//       query (Fmax <= next p: the overwhelmingly common case), the read's flush order is simply query by query, each
//       query's events in the order its record gives -- no pending list, no sorting.  A read that breaks the rule (it
//       repeats a k-mer prefix of its own) is handed to score_kernel untouched.  The lane first finds the (up to) two
//       taxa the read really comes from -- the first ones with a deep match -- and then replays ONLY their chains:
//       per query a level mask per taxon, |T_k| out of the record, two selects and two LDS adds per event.  What one
//       lane does rarely the wavefront does almost always, so nothing else happens in that loop; the contributions to
//       all other taxa (chance matches of short prefixes, ~150 per read against a 4e8-record index) are only counted.
//   score_other_kernel  one WAVEFRONT per read, a lane per query (coalesced 32-byte records): every contribution to
//       the other taxa becomes an 8-byte record in the read's staging row, in flush order (a prefix sum over the lanes'
//       counts places them), where row_merge_kernel resolves them per taxon.
//
// Neither kernel touches the profile tables: everything leaves as records, so both can be rerun.
// ------------------------------------------------------------------------------------------------

// One workgroup per tile.  (Tried: persistent workgroups, three per CU, each taking tiles gridDim.x apart, so that a tile's 1024
// scattered record stores -- what bounds the stage, tools/scatter_probe.hip -- drain while the workgroup computes its next
// tile: 71.7 ms against 63.7 at C2.  The hardware's own scheduling of fresh workgroups overlaps the phases better than a
// loop with a barrier at every tile's end.)
template <class Key, int NKT, int VAR = 0, int PARKX = 1>
__global__ __launch_bounds__(GTHREADS, PARKX == 1 ? 6 : 4) void group2_kernel(
    const Key *__restrict__ qKmer, const uint8_t *__restrict__ depth, const uint32_t *__restrict__ rep,
    const uint32_t *__restrict__ slotOf, uint32_t nQ, const uint32_t *__restrict__ tileNext, uint32_t nTiles,
    const typename KeyTraits<Key>::Meta *__restrict__ meta, const uint32_t *__restrict__ tax, uint32_t nIdx, int kHigh, int kLow,
    uint32_t *__restrict__ rec, uint32_t *__restrict__ pool, uint32_t poolCap, unsigned long long *__restrict__ poolCursor, int flags,
    uint32_t nTaxa, uint64_t *__restrict__ profKeys, uint32_t keyCap, unsigned long long *__restrict__ keyCursor, ProfLayout PL,
    uint64_t *__restrict__ cntAllHi, uint64_t *__restrict__ cntAllMid, uint64_t *__restrict__ cntAllLo,
    uint32_t *__restrict__ slowCount, uint32_t *__restrict__ slowList, uint32_t cellW, const uint32_t *__restrict__ tileIn)
{
    // (tileIn: the tiles a first launch listed -- PARKX > 1: those that only parked too many segments get a second chance here)
    group2_tile<Key, NKT, VAR, PARKX>(tileIn ? (tileIn[blockIdx.x] & 0x0FFFFFFFu) : blockIdx.x, qKmer, depth, rep, slotOf, nQ, tileNext, nTiles, meta, tax, nIdx, kHigh, kLow, rec, pool, poolCap, poolCursor, flags,
                          nTaxa, profKeys, keyCap, keyCursor, PL, cntAllHi, cntAllMid, cntAllLo, slowCount, slowList, cellW);
}

// Reads with LONG ROWS that keep the fast kernels' rule -- every group of a query is closed before the read's next matched
// query (Fmax <= p of the next) -- need no pending window: a query's events go to the row in the record's own order.  With a
// crowded index (conserved genes in clades of 50-200 taxa) a third of the reads are such: their staging rows would exceed
// RMAX records, score_main_kernel hands them over, and the general kernel replayed them event by event through its sorted
// window at ~1000 wavefront instructions per query.  Here: one wavefront per read, the read's row (nTaxa floats) in LDS, a
// query's segments dealt out to the lanes 64 at a time; a lane owns ITS segment's taxon for the query -- one LDS read of the
// cell, the query's events in flush order added in a register (a taxon's float chain only ever meets its own cell:
// Compare.hpp:528-530 adds hit by hit, the order that matters is per (read, taxon)), one LDS write.  A query in which a taxon
// may own several segments (REC_SPLIT) is taken segment by segment.  A read that breaks the rule is handed on to the general
// kernel (`hand`), its row cleared.  Narrow records (the profile is the group stage's).
static constexpr int SD_WAVES = 4;                                      // wavefronts (reads in flight) per workgroup
template <int NLV>
__global__ __launch_bounds__(64 * SD_WAVES) void score_dense_kernel(ScoreArgs A, uint32_t *__restrict__ hand, uint32_t *__restrict__ handCount)
{
    extern __shared__ float sRowDyn[];                                 // per wavefront: the row, nTaxa floats; then nTaxa bytes (REC_SPLIT queries)
    __shared__ EventTables evT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t perWave = A.nTaxa + (A.nTaxa + 3u) / 4u;            // (words)
    float *row = sRowDyn + (size_t)wv * perWave;
    uint32_t *sUni = reinterpret_cast<uint32_t *>(row + A.nTaxa);
    event_tables_init(evT);
    for (uint32_t tx = lane; tx < A.nTaxa; tx += 64) row[tx] = 0.0f;
    for (uint32_t x = lane; x < (A.nTaxa + 3u) / 4u; x += 64) sUni[x] = 0u;
    __syncthreads();
    const uint4 *rec4 = reinterpret_cast<const uint4 *>(A.rec);
    const uint64_t CQ = A.recCW / 4u;                                  // 16-byte words per cell
    for (uint32_t wi = blockIdx.x * SD_WAVES + (uint32_t)wv; wi < A.nList; wi += gridDim.x * SD_WAVES) {
        const uint32_t r = A.list[wi];
        const uint64_t o0 = A.kmerOff[r];
        const uint32_t cnt = (uint32_t)(A.kmerOff[r + 1] - o0);
        uint32_t prevF = 0;
        bool broke = false;
        // Three queries in flight: record j + 2 is being loaded, the pool words query j + 1 needs first -- its list's length if
        // 255 or more, the exact level sizes its events ask for, the first 64 segments -- are being loaded (their addresses come
        // from record j + 1 alone), query j is replayed.  One wavefront works on one read: without this every query was two
        // dependent round trips to memory with nothing else to do (71 ms for 700 000 reads of the crowded workload).
        struct Pre { uint32_t cnt, size, seg; };
        // (every load is issued on every path -- a word that is not needed is read from pool[0] -- so that the number of loads in
        // flight is the same whatever the records hold: the compiler can then wait for the OLDER ones only)
        auto prefetch = [&](const uint4 &ph, const uint4 &pb) -> Pre {
            const int d = (int)(ph.z & 31u);
            const uint32_t nf = ph.w & 255u, off = pb.w;
            const bool live = d != 0, sat = live && (ph.z & REC_SAT) != 0u;
            const uint32_t lv = ((ph.z >> 5) >> (3 * (lane & 7))) & 7u;
            const bool wantSize = sat && lane < d - A.kLow + 1 && (((ph.w >> 8) >> (3u * lv)) & 7u) == 7u;
            const uint32_t nInl = nf <= 4u ? nf : 3u;
            const bool wantSeg = live && (uint32_t)lane >= nInl && (uint32_t)lane < (nf == 255u ? 64u : nf);
            Pre P;
            P.cnt = A.pool[(live && nf == 255u) ? off : 0u];
            P.size = A.pool[wantSize ? off + 1u + (lv >> 1) : 0u];
            P.seg = A.pool[wantSeg ? off + 1u + (sat ? POOL_SIZES : 0u) + (uint32_t)lane - nInl : 0u];
            return P;
        };
        uint4 h = make_uint4(0, 0, 0, 0), b = h, h2 = h, b2 = h;
        if (cnt) { h = rec4[o0 * CQ]; b = rec4[o0 * CQ + 1]; }
        if (cnt) { const uint64_t s1 = o0 + (cnt > 1 ? 1u : 0u); h2 = rec4[s1 * CQ]; b2 = rec4[s1 * CQ + 1]; }
        Pre pre = cnt ? prefetch(h, b) : Pre{0u, 0u, 0u};
        for (uint32_t j = 0; j < cnt; ++j) {
            uint4 ch = h;
            const uint4 cb = b;
            const Pre cp = pre;
            h = h2; b = b2;
            { const uint64_t s2 = o0 + (j + 2 < cnt ? j + 2 : cnt - 1); h2 = rec4[s2 * CQ]; b2 = rec4[s2 * CQ + 1]; }   // (past the end: the last record again)
            pre = prefetch(h, b);
            // (the record is the whole wavefront's: its header words as scalars)
            ch.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.x); ch.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.y);
            ch.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.z); ch.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.w);
            const int d = (int)(ch.z & 31u);
            if (d == 0) continue;
            if (prevF > ch.x) { broke = true; break; }               // an earlier group of the read is still open: the general kernel's
            prevF = ch.y;
            const uint32_t order = (ch.z >> 5) & 0xFFFFFFu;
            const bool sat = (ch.z & REC_SAT) != 0u, split = (ch.z & REC_SPLIT) != 0u;
            uint32_t nseg = ch.w & 255u;
            const uint32_t nlev = ch.w >> 8;
            if (nseg == 255u) nseg = (uint32_t)__builtin_amdgcn_readfirstlane((int)cp.cnt);
            const uint32_t nInl = nseg <= 4u ? nseg : 3u;
            const uint32_t *sizes = A.pool + cb.w + 1u;              // (valid with REC_SAT)
            const uint32_t *more = sizes + (sat ? POOL_SIZES : 0u);  // (valid beyond nInl)
            const int nEv = d - A.kLow + 1;
            // lane e: level and score of the query's e-th event (Compare.hpp:923-924)
            int lvMine = 0; float sMine = 0.0f;
            if (lane < nEv) {
                lvMine = (int)((order >> (3 * lane)) & 7u);
                uint32_t n = (nlev >> (3 * lvMine)) & 7u;
                if (n == 7u) n = (cp.size >> (16 * (lvMine & 1))) & 0xFFFFu;   // "7 or more": the exact size is in the pool
                sMine = event_score(evT, A.kHigh - lvMine, n);
            }
            auto segAt = [&](uint32_t i) -> uint32_t {                // segment i of the query (the first 64 were fetched a query ahead)
                return i < nInl ? (i == 0u ? cb.x : i == 1u ? cb.y : i == 2u ? cb.z : cb.w) : (i < 64u ? cp.seg : more[i - nInl]);
            };
            // the query's events as scalars, once per query: level and score of event e (events beyond the query's last: level 0,
            // score 0 -- adding + 0.0f leaves a score as it is, so the replay below needs no branch)
            int lvE[NLV]; float sE[NLV];
#pragma unroll
            for (int e = 0; e < NLV; ++e) {
                lvE[e] = __builtin_amdgcn_readlane(lvMine, e);
                sE[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sMine), e));   // (lanes >= nEv hold 0.0f)
            }
            auto replay = [&](uint32_t levels, float acc) -> float {  // the query's events at the taxon's levels (bit lv = kHigh - k), in flush order
#pragma unroll
                for (int e = 0; e < NLV; ++e)                          // acc + s or acc + 0: one fused multiply-add by 1.0f or 0.0f (exact: the product is s or 0)
                    acc = __fmaf_rn((float)((levels >> lvE[e]) & 1u), sE[e], acc);
                return acc;
            };
            const int tap = A.forceHandOn;                            // (timing taps, KASA_DENSE_TAP: 1 no union sweep, 2 no replay, 4 one chunk per query)
            if (tap & 4) nseg = nseg < 64u ? nseg : 64u;
            // A long list's chunks of 64 segments: the loads of the next two chunks leave before this one is worked on (always
            // both, from a clamped place: the number of loads in flight is fixed), so that a list of 1400 taxa -- a k-mer every
            // genome shares -- is not 22 round trips to memory one after the other (43 of the kernel's 71 ms).
            auto forChunks = [&](auto f) {
                const uint32_t lastSeg = nseg - 1u;                    // (nseg >= 1: the query is matched)
                auto poolAt = [&](uint32_t i) -> uint32_t { const uint32_t x = i < lastSeg ? i : lastSeg; return more[(x < nInl ? nInl : x) - nInl]; };
                uint32_t s0 = segAt((uint32_t)lane < nseg ? (uint32_t)lane : 0u);     // chunk 0: fetched a query ahead
                uint32_t s1 = nseg > 64u ? poolAt(64u + (uint32_t)lane) : 0u, s2 = nseg > 128u ? poolAt(128u + (uint32_t)lane) : 0u;
                for (uint32_t c0 = 0; c0 < nseg; c0 += 64) {
                    const uint32_t cur = s0;
                    s0 = s1; s1 = s2;
                    if (c0 + 192u < nseg) s2 = poolAt(c0 + 192u + (uint32_t)lane);
                    const uint32_t i = c0 + (uint32_t)lane;
                    f(i, i < nseg, cur);
                }
            };
            if (split && !(tap & 1))
                forChunks([&](uint32_t, bool has, uint32_t sq) {
                    const uint32_t t = sq & SEG_TAX_MASK;
                    if (has) atomicOr(&sUni[t >> 2], seg_level_mask(sq, A.kHigh) << (8u * (t & 3u)));
                });
            if (split) LDS_WAVE_SYNC();
            forChunks([&](uint32_t, bool has, uint32_t sq) {
                const uint32_t t = sq & SEG_TAX_MASK;
                uint32_t levels = has ? seg_level_mask(sq, A.kHigh) : 0u;
                if (tap & 2) return;
                if (split && has && !(tap & 1)) {
                    const uint32_t sh = 8u * (t & 3u);
                    levels = (atomicAnd(&sUni[t >> 2], ~(0xFFu << sh)) >> sh) & 0xFFu;   // (0: another lane has taken the taxon)
                }
                const bool mine = has && levels != 0u;
                float acc = mine ? row[t] : 0.0f;
                acc = replay(levels, acc);
                if (mine) row[t] = acc;
            });
        }
        if (broke) {                                                  // hand the read on, leave no trace
            for (uint32_t tx = lane; tx < A.nTaxa; tx += 64) row[tx] = 0.0f;
            if (lane == 0) hand[atomicAdd(handCount, 1u)] = r;
            LDS_WAVE_SYNC();
            continue;
        }
        LDS_WAVE_SYNC();
        // the row (taxon ascending), as the general kernel emits it; the LDS row is cleared for the next read
        uint32_t m = 0;
        for (uint32_t b0 = 0; b0 < A.nTaxa; b0 += 64) { const uint32_t tx = b0 + lane; m += (uint32_t)__popcll(__ballot(tx < A.nTaxa && row[tx] > 0.0f)); }
        unsigned long long start = 0;
        if (lane == 0) {
            start = m ? atomicAdd(A.stCursor, (unsigned long long)m) : 0ull;
            const bool fits = start + m <= (unsigned long long)A.stCap;
            A.rowPos[r] = fits ? (uint32_t)start : 0u; A.rowLen[r] = fits ? m : 0u;
            if (!fits) start = ~0ull;
        }
        start = lane_value<0>(start);
        unsigned long long w = start;
        for (uint32_t b0 = 0; b0 < A.nTaxa; b0 += 64) {
            const uint32_t tx = b0 + lane;
            const float v = (tx < A.nTaxa) ? row[tx] : 0.0f;
            const unsigned long long mk = __ballot(v > 0.0f);
            if (v > 0.0f) {
                if (start != ~0ull) A.st[w + __popcll(mk & ((1ull << lane) - 1ull))] = make_uint2(tx, __float_as_uint(v));
                row[tx] = 0.0f;
            }
            w += __popcll(mk);
        }
        LDS_WAVE_SYNC();
    }
}

template <int RW> struct QueryRec {
    typedef RecTraits<RW> RT;
    uint32_t p, fmax, nseg, nlev, split;
    int d;
    unsigned __int128 order;
    uint32_t sg[RT::INL];
    uint32_t nInl, nMore;
    const uint32_t *more, *sizes;                                      // pool: further segments; exact level sizes (REC_SAT)
    // wide records carry no |T_k|: a kernel may build them once per query (level_sizes_*) and hang the table in here
    const uint32_t *tab = nullptr;
    int tabStride = 0;
    __device__ __forceinline__ void decode(const uint4 *rp, const uint32_t *__restrict__ pool)
    {
        uint4 v[RW / 4];
#pragma unroll
        for (int i = 0; i < RW / 4; ++i) v[i] = rp[i];
        decode_regs(v, pool);
    }
    __device__ __forceinline__ void decode_regs(const uint4 (&v)[RW / 4], const uint32_t *__restrict__ pool)
    {
        const uint4 h = v[0];
        p = h.x; fmax = h.y; d = (int)(h.z & 31u);
        if constexpr (RW == 8) {
            const uint4 b = v[1];
            order = (h.z >> 5) & 0xFFFFFFu; split = h.z & (REC_SPLIT | REC_SAT);
            nseg = h.w & 255u; nlev = h.w >> 8;
            sg[0] = b.x; sg[1] = b.y; sg[2] = b.z; sg[3] = b.w;
        } else {
            const uint4 o4 = v[1], s0 = v[2], s1 = v[3];
            order = ((unsigned __int128)o4.w << 96) | ((unsigned __int128)o4.z << 64) | ((unsigned __int128)o4.y << 32) | o4.x;
            split = h.z & REC_SPLIT; nseg = h.w; nlev = 0;
            sg[0] = s0.x; sg[1] = s0.y; sg[2] = s0.z; sg[3] = s0.w; sg[4] = s1.x; sg[5] = s1.y; sg[6] = s1.z; sg[7] = s1.w;
        }
        finish(pool);
    }
    __device__ __forceinline__ void finish(const uint32_t *__restrict__ pool)
    {
        nInl = nseg <= (uint32_t)RT::INL ? nseg : (uint32_t)RT::INL - 1u;
        nMore = nseg <= (uint32_t)RT::INL ? 0u : nseg - ((uint32_t)RT::INL - 1u);
        sizes = pool + sg[RT::INL - 1] + 1u;                           // valid with REC_SAT
        more = sizes + ((RW == 8 && (split & REC_SAT)) ? POOL_SIZES : 0u);   // valid when nMore > 0
        if (RW == 8 && nseg == 255u) { nseg = pool[sg[RT::INL - 1]]; nMore = nseg - ((uint32_t)RT::INL - 1u); }
        split &= REC_SPLIT;
    }
    __device__ __forceinline__ uint32_t set_size(int lv, int kHigh) const
    {
        uint32_t n = RW == 8 ? ((nlev >> (3 * lv)) & 7u) : 7u;
        if (RW == 8) return n < 7u ? n : ((sizes[lv >> 1] >> (16 * (lv & 1))) & 0xFFFFu);   // "7 or more": the exact size is in the pool
        if (tab) return lvp_get(tab, tabStride, lv);
        if (n == 7u) {                                               // not recorded: count
            const uint32_t k = (uint32_t)(kHigh - lv);
            n = 0;
#pragma unroll
            for (int q = 0; q < RT::INL; ++q) n += ((uint32_t)q < nInl && seg_covers(sg[q], k)) ? 1u : 0u;
            for (uint32_t q = 0; q < nMore; ++q) n += seg_covers(more[q], k) ? 1u : 0u;
        }
        return n;
    }
};

// per read, from score_main_kernel to score_other_kernel
struct MainOut { uint32_t tax0, tax1, otherAt, nOther; };             // otherAt: staging position of the other taxa's records

// Staging records one segment of another taxon yields (both fast kernels must agree): one segment record when it has at
// most two levels and every |T| fits its 3 bits, else one event record per level.  Wide records (RW = 16) carry no sizes.
template <int RW> __device__ __forceinline__ uint32_t seg_records(uint32_t levelMask, bool saturated)
{
    const uint32_t pc = (uint32_t)__popc(levelMask);
    return (RW == 8 && pc <= 2u && !saturated) ? (pc ? 1u : 0u) : pc;
}

// FB: bits of a hit counter.  16 by default; 8 when no read of the batch has more than 255 k-mers (narrow records): the LDS
// counters are what limits the resident wavefronts of this kernel, and it runs faster the more there are (26 ms at 16 per
// CU, 40 at 8).
// NLV: rows of the per-level LDS tables (the levels the launch can meet: 8 for narrow records; 25, or 19 for the default
// -k 25 7 of a 128-bit index -- the kernel's LDS footprint is what limits its resident wavefronts, and with 64-byte records
// and 16-bit fields it was down to six per CU).
// PF: the next line is loaded while this one is replayed.  Off for narrow records: the 32 registers that holds cost two of six
// resident wavefronts per SIMD, and those hide the latency better (22.0 -> 21.0 ms; debug flag 1024 runs the prefetching
// variant).  Wide records: the LDS tables limit the wavefronts anyway, prefetching wins (60 against 64 ms).
#ifndef KASA_MAIN16_WAVES
#define KASA_MAIN16_WAVES 1
#endif
template <int RW, bool PERREAD, int FB = 16, int NLV = RecTraits<RW>::LEVELS, bool PF = (RW != 8)>
__global__ __launch_bounds__(64, (RW == 8 && NLV <= 6 && FB == 8) ? 7 : (RW == 16 ? KASA_MAIN16_WAVES : 1)) void score_main_kernel(ScoreArgs A)
{
    typedef RecTraits<RW> RT;
    constexpr int NL = NLV, OB = RT::OBITS;
    constexpr bool GP = GpOf<RW>::v;
    // hit counters per |T|: 4 (wide records: 2) fields of FB bits
    typedef typename std::conditional<RW == 8, typename std::conditional<FB == 16, unsigned long long, uint32_t>::type,
                                      typename std::conditional<FB == 16, uint32_t, uint16_t>::type>::type Counter;
    constexpr uint32_t CNT_FIELDS = RW == 8 ? 4u : 2u;
    constexpr uint32_t FMASK = (1u << FB) - 1u;
    __shared__ Counter cnt[GP ? 1 : FTA * NL][GP ? 1 : 64];
    __shared__ float sTab[NL][8];                                    // score of one hit by (level, |T| < 8)
    __shared__ EventTables evT;
    __shared__ uint32_t sPB[65], sPPtr[64], sPT0[64], sPT1[64], sPM0[64], sPM1[64], sPRec[64], sPKey[64];   // pool segments of the current queries
    // wide records: |T_k| of the current query of every lane -- +1 / -1 at the ends of each segment's level range, then a
    // running sum over the levels (counting the segments again for every event cost seven times as much)
    __shared__ uint32_t sLvN[RW == 16 ? (NL + 3) / 2 : 1][64];
    event_tables_init(evT);
    const int lane = threadIdx.x;
    const int nK = A.kHigh - A.kLow + 1;
    for (int i = lane; i < NL * 8; i += 64) {
        const int lv = i >> 3, n = i & 7, k = A.kHigh - lv;
        sTab[lv][n] = (n > 0 && k >= 1) ? event_score(k, (uint32_t)n) : 0.0f;
    }
    __syncthreads();
    const int kPromote = (nK >= 3) ? A.kLow + 2 : A.kLow;           // shallow levels collect chance matches
    const uint32_t QS = A.recQS;                                     // log2(16-byte words per cell): narrow records may lie in 64-byte cells
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A.workCursor, 64u);          // persistent wavefronts take 64 reads at a time
        base = lane_value<0>(base);
        if (base >= A.nReads) break;
        const uint32_t r = base + lane;
        const bool active = r < A.nReads;
        bool fb = false;
        int na = 0;
        uint32_t mTax0 = 0xFFFFFFFFu, mTax1 = 0xFFFFFFFFu, nOther = 0, nKeys = 0;   // records / profile keys of the other taxa
        float mS0 = 0.0f, mS1 = 0.0f;
        if constexpr (!GP) for (int l2 = 0; l2 < FTA * nK; ++l2) cnt[(l2 / nK) * NL + (l2 % nK)][lane] = 0;
        uint64_t o0 = 0;
        uint32_t cnt0 = 0, rowCap = (uint32_t)RMAX;
        const uint4 *rp0 = reinterpret_cast<const uint4 *>(A.rec);
        if (active) {
            o0 = A.kmerOff[r];
            cnt0 = (uint32_t)(A.kmerOff[r + 1] - o0);
            // a LONG read has a long row because it has many k-mers, about one chance match of a short prefix each -- the row
            // merge streams such a row (row_merge_bitmap_kernel's LONG form); a row that is long because the taxon lists are
            // (a crowded index) still goes to score_dense_kernel
            if ((unsigned long long)A.rowPerQuery * cnt0 >= 2ull * RMAX) rowCap = (uint32_t)min((unsigned long long)A.rowPerQuery * cnt0, 0x0FFFFFFFull);
            if (cnt0 > (FB == 16 ? 60000u : 255u)) { fb = true; atomicAdd(&A.why[0], 1u); }   // the counters' fields
            rp0 = reinterpret_cast<const uint4 *>(A.rec) + (o0 << QS);
            // ---- A. the taxa that get the register slots: the first two with a deep match (segments come in descending
            // order of their last level: the search of a query ends at the first shallow one)
            for (uint32_t j = 0; j < cnt0 && na < FTA && !fb; ++j) {
                const uint4 *q4 = rp0 + ((size_t)j << QS);
                const uint4 h = q4[0];
                if ((int)(h.z & 31u) < kPromote) continue;                 // no segment reaches deeper than d (unmatched: d = 0)
                uint32_t sg[RT::INL];
                if constexpr (RW == 8) { const uint4 b4 = q4[1]; sg[0] = b4.x; sg[1] = b4.y; sg[2] = b4.z; sg[3] = b4.w; }
                else { const uint4 s0 = q4[2], s1 = q4[3]; sg[0] = s0.x; sg[1] = s0.y; sg[2] = s0.z; sg[3] = s0.w; sg[4] = s1.x; sg[5] = s1.y; sg[6] = s1.z; sg[7] = s1.w; }
                const uint32_t ns = RW == 8 ? (h.w & 255u) : h.w;
                const uint32_t nInl = ns <= (uint32_t)RT::INL ? ns : (uint32_t)RT::INL - 1u;
#pragma unroll
                for (int q = 0; q < RT::INL; ++q) {
                    const uint32_t t = sg[q] & SEG_TAX_MASK;
                    if ((uint32_t)q < nInl && (int)(sg[q] >> 27) >= kPromote && t != mTax0 && na < FTA) { if (na == 0) mTax0 = t; else mTax1 = t; ++na; }
                }
                if (ns > (uint32_t)RT::INL && na < FTA && (int)(sg[RT::INL - 2] >> 27) >= kPromote) {   // rare: the deep segments go on in the pool
                    const uint32_t *more = A.pool + sg[RT::INL - 1] + 1u + ((RW == 8 && (h.z & REC_SAT)) ? POOL_SIZES : 0u);
                    const uint32_t nAll = (RW == 8 && ns == 255u) ? A.pool[sg[RT::INL - 1]] : ns;
                    for (uint32_t q = 0; q + (uint32_t)(RT::INL - 1) < nAll && na < FTA; ++q) {
                        const uint32_t sq = more[q], t = sq & SEG_TAX_MASK;
                        if ((int)(sq >> 27) < kPromote) break;
                        if (t != mTax0) { if (na == 0) mTax0 = t; else mTax1 = t; ++na; }
                    }
                }
            }
        }
        // ---- B. their chains, query by query.  All lanes step together (the pool segments of the 64 current queries are
        // dealt out to the lanes: one dependent pool read per step instead of one per segment of the longest list).
        sPT0[lane] = mTax0; sPT1[lane] = mTax1;
        // A lane streams its read's records a whole 128-byte LINE at a time (4 narrow records, 2 wide ones), line-aligned:
        // the loads of a line leave together and nothing of it is fetched twice -- one record per step left every line to be
        // fetched again for the following steps unless a cache kept it (measured: 2-4 x the records' bytes from HBM).  A read
        // begins `ph` records into its first line; those steps are idle for the lane.  The next line is on its way while
        // this one is replayed.
        constexpr uint32_t LN = 32u / (uint32_t)RW, LW = LN * (RW / 4);        // records, 16-byte words of a line
        const uint32_t ph = (uint32_t)(o0 & (uint64_t)(LN - 1u));
        const uint4 *lp0 = rp0 - ((size_t)ph << QS);
        const uint32_t kEnd = (active && !fb) ? cnt0 + ph : 0u;               // this lane's steps: ph .. kEnd - 1
        uint32_t maxCnt = kEnd;
        for (int off = 32; off; off >>= 1) maxCnt = max(maxCnt, (uint32_t)__shfl_xor((int)maxCnt, off));
        {
            uint32_t prevF = 0;
            uint4 cur[LW], nxt[LW];
            auto loadLine = [&](uint32_t g) {                                  // records g LN .. g LN + LN - 1 of the lane's lines
#pragma unroll
                for (uint32_t r2 = 0; r2 < LN; ++r2) {
                    const uint32_t k = g * LN + r2;
                    const bool mineRec = !fb && k >= ph && k < kEnd;
#pragma unroll
                    for (uint32_t i = 0; i < (uint32_t)(RW / 4); ++i)
                        nxt[r2 * (RW / 4) + i] = mineRec ? lp0[((size_t)k << QS) + i] : make_uint4(0, 0, 0, 0);
                }
            };
            if (PF) loadLine(0);
            for (uint32_t g = 0; g * LN < maxCnt; ++g) {
            if (!PF) loadLine(g);
#pragma unroll
            for (uint32_t i = 0; i < LW; ++i) cur[i] = nxt[i];
            if (PF && (g + 1u) * LN < maxCnt) loadLine(g + 1u);
            for (uint32_t kk = 0; kk < LN; ++kk) {
                const uint32_t k = g * LN + kk, j = k - ph;
                const bool has = active && !fb && k >= ph && k < kEnd;
                uint4 v[RW / 4];
#pragma unroll
                for (int i = 0; i < RW / 4; ++i) v[i] = cur[i];
#pragma unroll
                for (uint32_t r2 = 1; r2 < LN; ++r2)
                    if (kk == r2) {                                            // (uniform)
#pragma unroll
                        for (int i = 0; i < RW / 4; ++i) v[i] = cur[r2 * (RW / 4) + i];
                    }
                if (has && ((o0 + j) & 63u) == 0u) A.otherOff64[(o0 + j) >> 6] = nOther;   // for score_other_kernel: the count so far at a wavefront's first slot
                bool live = has && (v[0].z & 31u) != 0u;
                if (!live) {
#pragma unroll
                    for (int i = 0; i < RW / 4; ++i) v[i] = make_uint4(0, 0, 0, 0);
                }
                QueryRec<RW> Q;
                Q.decode_regs(v, A.pool);
                if (PERREAD && live) {
                    if (prevF > Q.p) { fb = true; live = false; atomicAdd(&A.why[4], 1u); }   // an earlier group is still open here
                    else prevF = Q.fmax;
                }
                if (live && Q.nseg >= (1u << 13)) { fb = true; live = false; atomicAdd(&A.why[3], 1u); }
                const bool sat = RW == 8 && (v[0].z & REC_SPLIT) != 0u;        // then: one event record per level (seg_records)
                uint32_t mask0 = 0, mask1 = 0;
#pragma unroll
                for (int q = 0; q < RT::INL; ++q) {
                    if (!live || (uint32_t)q >= Q.nInl) continue;
                    const uint32_t t = Q.sg[q] & SEG_TAX_MASK, m = seg_level_mask(Q.sg[q], A.kHigh);
                    if (t == mTax0) mask0 |= m;
                    else if (t == mTax1) mask1 |= m;
                    else { nOther += seg_records<RW>(m, sat); if constexpr (!GP) nKeys += (uint32_t)__popc(m); }
                }
                if constexpr (RW == 16) {
                    for (uint32_t w = 0; w < lvp_rows(nK); ++w) sLvN[w][lane] = 0u;
#pragma unroll
                    for (int q = 0; q < RT::INL; ++q) {
                        if (!live || (uint32_t)q >= Q.nInl) continue;
                        const int la = A.kHigh - (int)(Q.sg[q] >> 27), lb = A.kHigh - (int)((Q.sg[q] >> 22) & 31u) + 1;
                        sLvN[la >> 1][lane] += lvp_unit(la);
                        sLvN[lb >> 1][lane] -= lvp_unit(lb);
                    }
                }
                const uint32_t nm = live ? Q.nMore : 0u;
                if (__ballot(nm != 0u) != 0ull) {
                    uint32_t incl = nm;
                    incl = wave_incl_sum(incl);
                    const uint32_t S = lane_value<63>(incl);
                    const unsigned long long satMask = __ballot(sat);
                    sPB[lane] = incl - nm;
                    if (lane == 63) sPB[64] = S;
                    sPPtr[lane] = (uint32_t)(Q.more - A.pool);
                    sPM0[lane] = 0u; sPM1[lane] = 0u; sPRec[lane] = 0u; sPKey[lane] = 0u;
                    LDS_WAVE_SYNC();
                    for (uint32_t b0 = 0; b0 < S; b0 += 64) {
                        const uint32_t i = b0 + lane;
                        if (i < S) {
                            uint32_t own = 0;
#pragma unroll
                            for (int step = 32; step; step >>= 1) if (sPB[own + step] <= i) own += step;
                            const uint32_t sq = A.pool[sPPtr[own] + (i - sPB[own])];
                            const uint32_t t = sq & SEG_TAX_MASK, m = seg_level_mask(sq, A.kHigh);
                            if constexpr (RW == 16) {
                                const int la = A.kHigh - (int)(sq >> 27), lb = A.kHigh - (int)((sq >> 22) & 31u) + 1;
                                atomicAdd(&sLvN[la >> 1][own], lvp_unit(la));
                                atomicSub(&sLvN[lb >> 1][own], lvp_unit(lb));
                            }
                            if (t == sPT0[own]) atomicOr(&sPM0[own], m);
                            else if (t == sPT1[own]) atomicOr(&sPM1[own], m);
                            else { atomicAdd(&sPRec[own], seg_records<RW>(m, ((satMask >> own) & 1ull) != 0ull)); if constexpr (!GP) atomicAdd(&sPKey[own], (uint32_t)__popc(m)); }
                        }
                    }
                    LDS_WAVE_SYNC();
                    mask0 |= sPM0[lane]; mask1 |= sPM1[lane]; nOther += sPRec[lane]; nKeys += sPKey[lane];
                    LDS_WAVE_SYNC();
                }
                if (live && nOther > rowCap) { fb = true; live = false; atomicAdd(&A.why[2], 1u); }   // a row longer than row_merge handles: the read is the general kernel's, no need to count on
                if ((mask0 | mask1) == 0u || !live) continue;
                if constexpr (RW == 16) {
                    LDS_WAVE_SYNC();
                    lvp_running(&sLvN[0][lane], 64, nK, [](int, uint32_t) {});
                    Q.tab = &sLvN[0][lane]; Q.tabStride = 64;
                }
                const int nEv = Q.d - A.kLow + 1;
                unsigned __int128 o = Q.order;
                for (int ev = 0; ev < nEv; ++ev, o >>= OB) {
                    const int lv = (int)((uint32_t)o & ((1u << OB) - 1u));
                    const uint32_t in0 = (mask0 >> lv) & 1u, in1 = (mask1 >> lv) & 1u;
                    if ((in0 | in1) == 0u) continue;
                    const uint32_t n = Q.set_size(lv, A.kHigh);
                    const float s = n < 8u ? sTab[lv][n] : event_score(evT, A.kHigh - lv, n);
                    if (PERREAD) {                                           // Compare.hpp:528-530; + 0.0f leaves a score as it is
                        mS0 = __fadd_rn(mS0, in0 ? s : 0.0f);
                        mS1 = __fadd_rn(mS1, in1 ? s : 0.0f);
                    }
                    if constexpr (!GP) {
                        if (n <= CNT_FIELDS) {
                            const Counter one = (Counter)1 << (FB * (n - 1u));
                            cnt[lv][lane] += in0 ? one : (Counter)0;
                            cnt[NL + lv][lane] += in1 ? one : (Counter)0;
                        } else { nOther += in0 + in1; nKeys += in0 + in1; }   // a profile record, written by score_other_kernel
                    }
                }
            }
            }
        }
        // ---- the read's staging row: final scores of the register taxa, their counters as profile records, then room
        // for the other taxa's records
        uint32_t nprof = 0;
        if (!GP && active && !fb)
            for (int e = 0; e < na; ++e)
                for (int lv = 0; lv < nK; ++lv) {
                    const unsigned long long pk = cnt[e * NL + lv][lane];
                    for (uint32_t q = 0; q < CNT_FIELDS; ++q) nprof += ((pk >> (FB * q)) & FMASK) != 0 ? 1u : 0u;
                }
        const uint32_t nFinal = PERREAD ? (uint32_t)na : 0u;
        if (active && !fb && nFinal + nprof + nOther > rowCap) { fb = true; atomicAdd(&A.why[2], 1u); }   // longer than row_merge handles
        const uint32_t m = (active && !fb) ? nFinal + nprof + nOther : 0u;
        uint32_t incl = m;
        incl = wave_incl_sum(incl);
        const uint32_t total = lane_value<63>(incl);
        unsigned long long start = 0;
        if (lane == 0 && total) start = atomicAdd(A.stCursor, (unsigned long long)total);   // 64-bit: the host sees the true demand
        start = lane_value<0>(start);
        start += incl - m;
        // ... and its profile keys: one per counter record and per event of the other taxa
        const uint32_t mk = (active && !fb) ? nprof + nKeys : 0u;
        uint32_t inclK = mk;
        inclK = wave_incl_sum(inclK);
        const uint32_t totalK = lane_value<63>(inclK);
        unsigned long long startK = 0;
        if (lane == 0 && totalK) startK = atomicAdd(A.keyCursor, (unsigned long long)totalK);
        startK = lane_value<0>(startK);
        const bool fits = start - (incl - m) + total <= (unsigned long long)A.stCap && startK + totalK <= (unsigned long long)A.keyCap;
        startK += inclK - mk;
        if (active) {
            MainOut mo{mTax0, mTax1, 0u, 0u};
            if (!fb) {
                A.rowPos[r] = fits ? (uint32_t)start : 0u; A.rowLen[r] = fits ? (m | (m ? ROW_MERGE : 0u)) : 0u;
                A.rowKey[r] = fits ? (uint32_t)startK : 0u;
                if (fits) {
                    uint32_t w = (uint32_t)start;
                    if (PERREAD && na > 0) A.st[w++] = make_uint2(mTax0 | RK_FINAL, __float_as_uint(mS0));
                    if (PERREAD && na > 1) A.st[w++] = make_uint2(mTax1 | RK_FINAL, __float_as_uint(mS1));
                    for (int e = 0; !GP && e < na; ++e) {
                        const uint32_t t = (e == 0) ? mTax0 : mTax1;
                        for (int lv = 0; lv < nK; ++lv) {
                            const unsigned long long pk = cnt[e * NL + lv][lane];
                            for (uint32_t q = 0; q < CNT_FIELDS; ++q) {
                                const uint32_t cq = (uint32_t)((pk >> (FB * q)) & FMASK);
                                if (cq) A.st[w++] = make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | RK_PROFILE, ((q + 1) << 16) | cq);
                            }
                        }
                    }
                    mo.otherAt = w; mo.nOther = nOther;
                }
            }
            reinterpret_cast<uint4 *>(A.mainOut)[r] = make_uint4(mo.tax0, mo.tax1, mo.otherAt, mo.nOther);
        }
        const unsigned long long fbMask = __ballot(active && fb);
        if (fbMask) {
            uint32_t fbBase = 0;
            if (lane == 0) fbBase = atomicAdd(A.fbCount, (uint32_t)__popcll(fbMask));
            fbBase = lane_value<0>(fbBase);
            if (active && fb) A.fbList[fbBase + __popcll(fbMask & ((1ull << lane) - 1ull))] = r;
        }
    }
}

// The other taxa's contributions of the reads score_main_kernel kept: one THREAD per query (slot), a wavefront = 64
// consecutive slots (coalesced 32-byte records, every lane busy, reads may begin and end inside a wavefront).  A lane
// counts the staging records its query yields -- one per segment, as a rule (seg_records) -- and a segmented prefix sum
// (restarting at every read's first slot) places them behind the read's `otherAt`; for the read that began before the
// wavefront, score_main_kernel left the running count at the wavefront's first slot (otherOff64).  So a row holds the
// records in the read's flush order: query by query (inside a query only the order among one taxon's records matters,
// and a taxon owns one segment unless the query is marked REC_SPLIT).  row_merge_kernel expands segment records.
template <int RW, bool PERREAD>
__global__ __launch_bounds__(256) void score_other_kernel(ScoreArgs A)
{
    typedef RecTraits<RW> RT;
    constexpr int OB = RT::OBITS;
    constexpr bool GP = GpOf<RW>::v;
    constexpr uint32_t CNT_FIELDS = RW == 8 ? 4u : 2u;
    typedef typename std::conditional<RW == 8, uint32_t, unsigned __int128>::type Order;
    const int lane = threadIdx.x & 63;
    const uint32_t kindOther = PERREAD ? 0u : RK_PROFILE;
    const uint32_t stride = gridDim.x * 256u;
    const double readsPerSlot = (double)A.nReads / (double)A.nQ;
    const uint32_t nQup = (A.nQ + 63u) & ~63u;                             // whole wavefronts take part in the prefix sums
    // A wavefront's records leave through LDS: a lane's 8-byte stores would reach the row as ~50 partial-line write requests
    // per wavefront (measured: 4/5 of the kernel); staged, consecutive lanes write consecutive records.
    constexpr uint32_t STAGE = 512;                                        // records a wavefront stages; larger (rare) batches are written directly
    __shared__ uint2 sRec[4][STAGE];
    __shared__ uint32_t sAt[4][STAGE];
    __shared__ uint32_t sLvN[RW == 16 ? (RT::LEVELS + 3) / 2 : 1][256];  // wide records: |T_k| of this thread's query (QueryRec::tab), two levels per word
    const int wv = threadIdx.x >> 6;
    for (uint32_t slot = blockIdx.x * 256u + threadIdx.x; slot < nQup; slot += stride) {
        const bool inRange = slot < A.nQ;
        uint4 cur[RW / 4];
#pragma unroll
        for (int i = 0; i < RW / 4; ++i) cur[i] = make_uint4(0, 0, 0, 0);
        uint32_t r = 0;
        uint64_t readStart = 0;
        uint4 mo = make_uint4(0, 0, 0, 0);
        if (inRange) {
#pragma unroll
            for (int i = 0; i < RW / 4; ++i) cur[i] = reinterpret_cast<const uint4 *>(A.rec)[(size_t)slot * (A.recCW / 4u) + i];
            // the read of this slot: reads are mostly equally long, so the proportional guess is right; else search
            r = (uint32_t)((double)slot * readsPerSlot);
            if (r >= A.nReads) r = A.nReads - 1u;
            readStart = A.kmerOff[r];
            if (!(readStart <= slot && slot < A.kmerOff[r + 1])) {
                uint32_t lo = 0, hi = A.nReads;                                // last r with kmerOff[r] <= slot
                while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (A.kmerOff[mid] <= slot) lo = mid; else hi = mid; }
                r = lo;
                readStart = A.kmerOff[r];
            }
            mo = reinterpret_cast<const uint4 *>(A.mainOut)[r];
        }
        const bool live = inRange && mo.w != 0u && (cur[0].z & 31u) != 0u;     // a matched query of a read the fast path kept
        const uint32_t mTax0 = mo.x, mTax1 = mo.y;
        QueryRec<RW> Q;
        Q.decode_regs(cur, A.pool);
        if (!live) { Q.d = 0; Q.nInl = 0; Q.nMore = 0; Q.split = 0; }
        const bool sat = RW == 8 && (cur[0].z & REC_SPLIT) != 0u;   // then: one event record per level (seg_records)
        const int nEv = Q.d ? Q.d - A.kLow + 1 : 0;
        // the pool segments of the query, the first four in registers (their loads go out together)
        uint32_t xs[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) if ((uint32_t)i < Q.nMore) xs[i] = Q.more[i];
        auto extra = [&](uint32_t q) -> uint32_t { return q == 0 ? xs[0] : q == 1 ? xs[1] : q == 2 ? xs[2] : q == 3 ? xs[3] : Q.more[q]; };
        if constexpr (RW == 16) {
            if (live) {                                                        // + 1 / - 1 at the ends of every segment's level range, running sum
                const int nKl = A.kHigh - A.kLow + 1, me = (int)threadIdx.x;
                for (uint32_t w = 0; w < lvp_rows(nKl); ++w) sLvN[w][me] = 0u;
                auto mark = [&](uint32_t sq) {
                    const int la = A.kHigh - (int)(sq >> 27), lb = A.kHigh - (int)((sq >> 22) & 31u) + 1;
                    sLvN[la >> 1][me] += lvp_unit(la);
                    sLvN[lb >> 1][me] -= lvp_unit(lb);
                };
#pragma unroll
                for (int q = 0; q < RT::INL; ++q) {
                    if ((uint32_t)q >= Q.nInl) continue;
                    mark(Q.sg[q]);
                }
                for (uint32_t q = 0; q < Q.nMore; ++q) mark(extra(q));
                lvp_running(&sLvN[0][me], 256, nKl, [](int, uint32_t) {});
                Q.tab = &sLvN[0][me]; Q.tabStride = 256;
            }
        }
        // levels with |T| > CNT_FIELDS: there the register taxa leave profile records instead of counting in LDS
        uint32_t bigLv = 0;
        if constexpr (RW == 8) {
            const uint32_t x = Q.nlev, f = (x >> 2) & ((x >> 1) | x) & 0x249249u;      // field >= 5, one bit per 3-bit field
            bigLv = (f & 1u) | ((f >> 2) & 2u) | ((f >> 4) & 4u) | ((f >> 6) & 8u) | ((f >> 8) & 16u) | ((f >> 10) & 32u) | ((f >> 12) & 64u) | ((f >> 14) & 128u);
        } else {
            for (int lv = 0; lv < nEv; ++lv) { const int l2 = A.kHigh - A.kLow - lv; if (Q.set_size(l2, A.kHigh) > CNT_FIELDS) bigLv |= 1u << l2; }
        }
        // levels at which a segment leaves records (all of them: another taxon; those with a large |T|: register taxon)
        // and how many records that makes
        auto emitMask = [&](uint32_t sq, bool valid) -> uint32_t {
            const uint32_t t = sq & SEG_TAX_MASK, m = valid ? seg_level_mask(sq, A.kHigh) : 0u;
            return (t != mTax0 && t != mTax1) ? m : (GP ? 0u : (m & bigLv));
        };
        auto records = [&](uint32_t sq, uint32_t m) -> uint32_t {
            const uint32_t t = sq & SEG_TAX_MASK;
            return (t != mTax0 && t != mTax1) ? seg_records<RW>(m, sat) : (uint32_t)__popc(m);
        };
        uint32_t em[RT::INL];
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < RT::INL; ++q) { em[q] = emitMask(Q.sg[q], (uint32_t)q < Q.nInl); mine += records(Q.sg[q], em[q]); }
        for (uint32_t q = 0; q < Q.nMore; ++q) { const uint32_t sq = extra(q); mine += records(sq, emitMask(sq, true)); }
        // segmented inclusive prefix sum over the wavefront: a segment starts at every read's first slot
        const bool head = inRange && (uint64_t)slot == readStart;
        uint32_t incl = mine;
        bool started = head;                                                   // a read starts at or before this lane, inside the wavefront
        wave_seg_incl_sum(incl, started);
        // place in the wavefront's staging area: a plain prefix sum
        uint32_t lincl = mine;
        lincl = wave_incl_sum(lincl);
        const uint32_t total = lane_value<63>(lincl);
        if (total == 0u) continue;                                           // uniform
        const bool staged = total <= STAGE;
        uint32_t lp = lincl - mine;
        // reads that began before the wavefront continue from the count score_main_kernel left at the wavefront's first slot
        uint32_t w = mine ? mo.z + incl - mine + (started ? 0u : A.otherOff64[slot >> 6]) : 0u;
        auto emit = [&](uint2 rec) {
            if (staged) { sRec[wv][lp] = rec; sAt[wv][lp] = w; ++lp; ++w; }
            else A.st[w++] = rec;
        };
        // position of level lv in the query's flush order (three bits per event: the field that equals lv, found without a loop)
        auto posOf = [&](int lv) -> int {
            if constexpr (RW == 8) {
                const uint32_t x = ((uint32_t)Q.order ^ ((uint32_t)lv * 0x249249u)) | (nEv < 8 ? (0xFFFFFFFFu << (3 * nEv)) : 0u);
                const uint32_t z = ~(x | (x >> 1) | (x >> 2)) & 0x249249u;       // bit 3 i set iff field i is zero
                return (__ffs((int)z) - 1) / 3;
            } else {
                Order o = (Order)Q.order;
                int pos = 0;
                for (int ev = 0; ev < nEv; ++ev, o >>= OB) if ((int)((uint32_t)o & ((1u << OB) - 1u)) == lv) pos = ev;
                return pos;
            }
        };
        auto putEvent = [&](uint32_t sq, int lv) {
            const uint32_t t = sq & SEG_TAX_MASK;
            const uint32_t kind = (t == mTax0 || t == mTax1) ? RK_PROFILE : kindOther;
            emit(make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | kind, (Q.set_size(lv, A.kHigh) << 16) | 1u));
        };
        auto putSeg = [&](uint32_t sq, uint32_t m) {
            if (m == 0u) return;
            const uint32_t t = sq & SEG_TAX_MASK;
            const uint32_t pc = (uint32_t)__popc(m);
            if (pc == 1u && (RW != 8 || t == mTax0 || t == mTax1 || sat)) { putEvent(sq, __ffs((int)m) - 1); return; }   // one event, one record
            if (RW == 8 && t != mTax0 && t != mTax1 && pc <= 2u && !sat) {   // one segment record (seg_records)
                const int lvLo = __ffs((int)m) - 1, lvHi = 31 - __clz((int)m);   // lvLo = the deeper level (larger k)
                const bool desc = pc == 2u && posOf(lvLo) < posOf(lvHi);          // the larger k flushes first
                emit(make_uint2(t | ((uint32_t)(A.kHigh - lvHi) << 20) | (desc ? RK_SEG_DESC : 0u) | RK_SEG,
                                (uint32_t)(A.kHigh - lvLo) | (Q.set_size(lvHi, A.kHigh) << 5) | (Q.set_size(lvLo, A.kHigh) << 18)));
                return;
            }
            Order o = (Order)Q.order;                                        // one event record per level, in flush order
            for (int ev = 0; ev < nEv; ++ev, o >>= OB) {
                const int lv = (int)((uint32_t)o & ((1u << OB) - 1u));
                if ((m >> lv) & 1u) putEvent(sq, lv);
            }
        };
        if (mine == 0u) {}
        else if (Q.split) {
            // a taxon may own several segments: its records must follow the query's flush order across them -- event by event
            Order o = (Order)Q.order;
            for (int ev = 0; ev < nEv; ++ev, o >>= OB) {
                const int lv = (int)((uint32_t)o & ((1u << OB) - 1u));
#pragma unroll
                for (int q = 0; q < RT::INL; ++q) if ((em[q] >> lv) & 1u) putEvent(Q.sg[q], lv);
                for (uint32_t q = 0; q < Q.nMore; ++q) { const uint32_t sq = extra(q); if ((emitMask(sq, true) >> lv) & 1u) putEvent(sq, lv); }
            }
        } else {
#pragma unroll
            for (int q = 0; q < RT::INL; ++q) putSeg(Q.sg[q], em[q]);
            for (uint32_t q = 0; q < Q.nMore; ++q) { const uint32_t sq = extra(q); putSeg(sq, emitMask(sq, true)); }
        }
        if (staged) {
            LDS_WAVE_SYNC();
            for (uint32_t x = lane; x < total; x += 64) A.st[sAt[wv][x]] = sRec[wv][x];
            LDS_WAVE_SYNC();
        }
    }
}

// The same for 32-byte records, flattened: the work items of a wavefront are the SEGMENTS of its 64 queries, dealt out
// evenly to the lanes (64 items per round) -- a query has 1 to hundreds of segments, and a lane that walks its own list
// keeps the other 63 waiting (the wavefront pays the longest list, and every rare case, each time).  The queries' fields
// an item needs sit in LDS; the owner of item i is found by bisection over the prefix sums of the segment counts.  An
// item knows how many records it yields (seg_records), so a segmented prefix sum over the items -- restarting where a
// new read begins -- gives each record its place in the read's row, and neighbouring lanes write neighbouring records.
// Queries marked REC_SPLIT (rare) are one item: their owner lane writes their records event by event afterwards.
template <bool PERREAD>
__global__ __launch_bounds__(256, 8) void score_other_flat_kernel(ScoreArgs A)
{
    constexpr int WV = 4;
    constexpr bool GP = GpOf<8>::v;
    __shared__ uint32_t sBase[WV][65];                                    // exclusive prefix sums of the queries' item counts
    __shared__ uint32_t sSg[WV][4][64];                                    // the inline segments that leave records, compacted
    __shared__ uint32_t sW2[WV][64], sW3[WV][64], sT0[WV][64], sT1[WV][64], sRow[WV][64], sBig[WV][64], sSplit[WV][64], sPool[WV][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t kindOther = PERREAD ? 0u : RK_PROFILE;
    const uint32_t stride = gridDim.x * 256u;
    const double readsPerSlot = (double)A.nReads / (double)A.nQ;
    const uint32_t nQup = (A.nQ + 63u) & ~63u;
    const unsigned long long upTo = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);   // lanes 0..lane
    for (uint32_t slot = blockIdx.x * 256u + threadIdx.x; slot < nQup; slot += stride) {
        const bool inRange = slot < A.nQ;
        uint4 cur[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        uint32_t r = 0;
        uint64_t readStart = 0;
        uint4 mo = make_uint4(0, 0, 0, 0);
        if (inRange) {
            cur[0] = reinterpret_cast<const uint4 *>(A.rec)[(size_t)slot << A.recQS];
            cur[1] = reinterpret_cast<const uint4 *>(A.rec)[((size_t)slot << A.recQS) + 1];
            r = (uint32_t)((double)slot * readsPerSlot);                       // reads are mostly equally long: the guess is right, else search
            if (r >= A.nReads) r = A.nReads - 1u;
            readStart = A.kmerOff[r];
            if (!(readStart <= slot && slot < A.kmerOff[r + 1])) {
                uint32_t lo = 0, hi = A.nReads;
                while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (A.kmerOff[mid] <= slot) lo = mid; else hi = mid; }
                r = lo;
                readStart = A.kmerOff[r];
            }
            mo = reinterpret_cast<const uint4 *>(A.mainOut)[r];
        }
        const bool live = inRange && mo.w != 0u && (cur[0].z & 31u) != 0u;     // a matched query of a read the fast path kept
        const uint32_t mTax0 = mo.x, mTax1 = mo.y;
        QueryRec<8> Q;
        Q.decode_regs(cur, A.pool);
        if (!live) { Q.d = 0; Q.nInl = 0; Q.nMore = 0; Q.split = 0; Q.nseg = 0; }
        const bool split = Q.split != 0u;
        const int nEvMine = Q.d ? Q.d - A.kLow + 1 : 0;
        // levels with |T| > 4: there the register taxa leave profile records instead of counting in LDS (score_main_kernel)
        const uint32_t xl = Q.nlev, fl = (xl >> 2) & ((xl >> 1) | xl) & 0x249249u;   // field >= 5, one bit per 3-bit field
        const uint32_t bigLv = (fl & 1u) | ((fl >> 2) & 2u) | ((fl >> 4) & 4u) | ((fl >> 6) & 8u) | ((fl >> 8) & 16u) | ((fl >> 10) & 32u) | ((fl >> 12) & 64u) | ((fl >> 14) & 128u);
        auto emitMask = [&](uint32_t sq) -> uint32_t {
            const uint32_t t = sq & SEG_TAX_MASK, m = seg_level_mask(sq, A.kHigh);
            return (t != mTax0 && t != mTax1) ? m : (GP ? 0u : (m & bigLv));
        };
        uint32_t mineSplit = 0;                                                // a split query's records: one per (segment, level)
        if (__ballot(split) != 0ull && split) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if ((uint32_t)q < Q.nInl) mineSplit += (uint32_t)__popc(emitMask(Q.sg[q]));
            for (uint32_t q = 0; q < Q.nMore; ++q) mineSplit += (uint32_t)__popc(emitMask(Q.more[q]));
        }
        // items: the inline segments that leave records (most belong to the register taxa and do not) and all pool segments
        uint32_t emInl = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if ((uint32_t)q < Q.nInl && emitMask(Q.sg[q]) != 0u) emInl |= 1u << q;
        const uint32_t nEmInl = (uint32_t)__popc(emInl);
        const uint32_t nFlat = split ? (mineSplit ? 1u : 0u) : nEmInl + Q.nMore;
        uint32_t incl = nFlat;
        incl = wave_incl_sum(incl);
        const uint32_t S = lane_value<63>(incl);
        if (S == 0u) continue;                                                 // uniform
        const bool head = inRange && (uint64_t)slot == readStart;
        const unsigned long long H = __ballot(head);
        const bool started = (H & upTo) != 0ull;                               // a read starts at or before this lane, inside the wavefront
        sBase[wv][lane] = incl - nFlat;
        if (lane == 63) sBase[wv][64] = S;
#pragma unroll
        for (int q = 0; q < 4; ++q) if ((emInl >> q) & 1u) sSg[wv][__popc(emInl & ((1u << q) - 1u))][lane] = Q.sg[q];
        sPool[wv][lane] = Q.sg[3];                                             // pool block of the query (when it has one)
        sW2[wv][lane] = cur[0].z; sW3[wv][lane] = (cur[0].w & ~255u) | (nEmInl << 4) | (Q.nMore ? 1u : 0u);   // low byte: items in LDS; has a pool block
        sT0[wv][lane] = mTax0; sT1[wv][lane] = mTax1; sBig[wv][lane] = bigLv;
        // reads that began before the wavefront continue from the count score_main_kernel left at the wavefront's first slot
        sRow[wv][lane] = live ? mo.z + (started ? 0u : A.otherOff64[slot >> 6]) : 0u;
        sSplit[wv][lane] = mineSplit;
        LDS_WAVE_SYNC();
        uint32_t carry = 0;
        int prevOwnCarry = -1;
        for (uint32_t b0 = 0; b0 < S; b0 += 64) {
            const uint32_t i = b0 + lane;
            const bool act = i < S;
            uint32_t own = 0;                                                  // the last query whose items start at or before i
#pragma unroll
            for (int step = 32; step; step >>= 1) if (sBase[wv][own + step] <= i) own += step;
            if (!act) own = 63u;
            const uint32_t idx = i - sBase[wv][own];
            const uint32_t w2 = sW2[wv][own], w3 = sW3[wv][own], poolAt = sPool[wv][own];
            const bool isSplit = (w2 & REC_SPLIT) != 0u;
            uint32_t sq = 0;
            const bool seg = act && !isSplit;
            if (seg) {
                const uint32_t nInl = (w3 >> 4) & 7u;
                if (idx < nInl) sq = sSg[wv][idx][own];
                else sq = A.pool[poolAt + 1u + ((w2 & REC_SAT) ? POOL_SIZES : 0u) + idx - nInl];
            }
            const uint32_t t = sq & SEG_TAX_MASK;
            const bool isMain = t == sT0[wv][own] || t == sT1[wv][own];
            const uint32_t m = seg ? seg_level_mask(sq, A.kHigh) : 0u;
            const uint32_t em = isMain ? (GP ? 0u : (m & sBig[wv][own])) : m;
            const uint32_t pc = (uint32_t)__popc(em);
            const bool segRec = !isMain && pc <= 2u;                           // one segment record (seg_records)
            uint32_t c = segRec ? (pc ? 1u : 0u) : pc;
            if (act && isSplit) c = sSplit[wv][own];
            // does a read begin between the previous item's query and this one's?  Then the count restarts here.
            const int prevOwn = lane_before((int)own, prevOwnCarry);
            const unsigned long long toOwn = own == 63u ? ~0ull : ((2ull << own) - 1ull);
            const unsigned long long toPrev = prevOwn < 0 ? 0ull : (prevOwn == 63 ? ~0ull : ((2ull << prevOwn) - 1ull));
            bool f = act && (H & toOwn & ~toPrev) != 0ull;
            uint32_t v = c;
            wave_seg_incl_sum(v, f);
            uint32_t w = sRow[wv][own] + v - c + (f ? 0u : carry);
            carry = lane_value<63>(v) + ((int)lane_value<63>((uint32_t)f) ? 0u : carry);
            prevOwnCarry = (int)lane_value<63>((uint32_t)own);
            const int nEv = (int)(w2 & 31u) - A.kLow + 1;
            const uint32_t order = (w2 >> 5) & 0xFFFFFFu;
            auto sizeOf = [&](int lv) -> uint32_t {
                const uint32_t n = (w3 >> (8 + 3 * lv)) & 7u;
                return n < 7u ? n : ((A.pool[poolAt + 1u + (uint32_t)(lv >> 1)] >> (16 * (lv & 1))) & 0xFFFFu);
            };
            if (act && c) {
                if (isSplit) sSplit[wv][own] = w;                             // its owner writes them below
                else if (segRec) {
                    const int lvLo = __ffs((int)em) - 1, lvHi = 31 - __clz((int)em);   // lvLo = the deeper level (larger k)
                    // position of a level in the query's flush order (three bits per event): the field that equals lv
                    auto posOf = [&](int lv) -> int {
                        const uint32_t x = (order ^ ((uint32_t)lv * 0x249249u)) | (nEv < 8 ? (0xFFFFFFFFu << (3 * nEv)) : 0u);
                        const uint32_t z = ~(x | (x >> 1) | (x >> 2)) & 0x249249u;
                        return (__ffs((int)z) - 1) / 3;
                    };
                    const bool desc = pc == 2u && posOf(lvLo) < posOf(lvHi);   // the larger k flushes first
                    A.st[w] = make_uint2(t | ((uint32_t)(A.kHigh - lvHi) << 20) | (desc ? RK_SEG_DESC : 0u) | RK_SEG,
                                         (uint32_t)(A.kHigh - lvLo) | (sizeOf(lvHi) << 5) | (sizeOf(lvLo) << 18));
                }
            }
            const bool multi = act && c != 0u && !isSplit && !segRec;          // one event record per level, in flush order
            if (__ballot(multi) != 0ull && multi) {
                const uint32_t kind = isMain ? RK_PROFILE : kindOther;
                uint32_t o = order;
                for (int ev = 0; ev < nEv; ++ev, o >>= 3) {
                    const int lv = (int)(o & 7u);
                    if ((em >> lv) & 1u) A.st[w++] = make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | kind, (sizeOf(lv) << 16) | 1u);
                }
            }
        }
        if (__ballot(split && mineSplit) != 0ull) {
            LDS_WAVE_SYNC();
            if (split && mineSplit) {
                // a taxon may own several segments: its records must follow the query's flush order across them -- event by event
                uint32_t w = sSplit[wv][lane];
                uint32_t o = (uint32_t)Q.order;
                auto putEvent = [&](uint32_t sq, int lv) {
                    const uint32_t t = sq & SEG_TAX_MASK;
                    const uint32_t kind = (t == mTax0 || t == mTax1) ? RK_PROFILE : kindOther;
                    A.st[w++] = make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | kind, (Q.set_size(lv, A.kHigh) << 16) | 1u);
                };
                for (int ev = 0; ev < nEvMine; ++ev, o >>= 3) {
                    const int lv = (int)(o & 7u);
#pragma unroll
                    for (int q = 0; q < 4; ++q) if ((uint32_t)q < Q.nInl && ((emitMask(Q.sg[q]) >> lv) & 1u)) putEvent(Q.sg[q], lv);
                    for (uint32_t q = 0; q < Q.nMore; ++q) { const uint32_t sq = Q.more[q]; if ((emitMask(sq) >> lv) & 1u) putEvent(sq, lv); }
                }
            }
        }
        LDS_WAVE_SYNC();
    }
}

// The flattened kernel for 64-byte records (up to 25 levels).  Wide records carry no |T_k|, so there are two sweeps over
// the segments of a wavefront's 64 queries: the first adds +1 / -1 at the ends of every segment's level range into the
// owner's column of an LDS table (a running sum over the levels then gives |T_k|, and the crowded levels of the register
// taxa), the second emits: every record of a wide query is an event record, a segment's levels leave in the query's flush
// order (5 bits per event in the record's 128-bit order field).
template <bool PERREAD, int NLW = RecTraits<16>::LEVELS>             // NLW: rows of the |T_k| table (19 for the default -k 25 7: less LDS, more wavefronts)
__global__ __launch_bounds__(128) void score_other_flat16_kernel(ScoreArgs A)
{
    typedef RecTraits<16> RT;
    constexpr int WV = 2, INL = RT::INL;
    constexpr bool GP = GpOf<16>::v;
    constexpr uint32_t CNT_FIELDS = 2u;
    __shared__ uint32_t sBase[WV][65];
    __shared__ uint32_t sSg[WV][INL][64];                                  // inline segments (sweep 1: all; sweep 2: those that leave records)
    __shared__ uint32_t sOrd[WV][4][64];                                   // flush order, 5 bits per event
    __shared__ uint32_t sW2[WV][64], sT0[WV][64], sT1[WV][64], sRow[WV][64], sBig[WV][64], sSplit[WV][64], sPool[WV][64], sNInl[WV][64];
    __shared__ uint32_t sLvN[WV][(NLW + 3) / 2][64];                       // |T_k| of every lane's query (QueryRec::tab), two levels per word
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t kindOther = PERREAD ? 0u : RK_PROFILE;
    const uint32_t stride = gridDim.x * 128u;
    const double readsPerSlot = (double)A.nReads / (double)A.nQ;
    const uint32_t nQup = (A.nQ + 63u) & ~63u;
    const unsigned long long upTo = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int nK = A.kHigh - A.kLow + 1;
    for (uint32_t slot = blockIdx.x * 128u + threadIdx.x; slot < nQup; slot += stride) {
        const bool inRange = slot < A.nQ;
        uint4 cur[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = make_uint4(0, 0, 0, 0);
        uint32_t r = 0;
        uint64_t readStart = 0;
        uint4 mo = make_uint4(0, 0, 0, 0);
        if (inRange) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[i] = reinterpret_cast<const uint4 *>(A.rec)[(size_t)slot * 4 + i];
            r = (uint32_t)((double)slot * readsPerSlot);
            if (r >= A.nReads) r = A.nReads - 1u;
            readStart = A.kmerOff[r];
            if (!(readStart <= slot && slot < A.kmerOff[r + 1])) {
                uint32_t lo = 0, hi = A.nReads;
                while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (A.kmerOff[mid] <= slot) lo = mid; else hi = mid; }
                r = lo;
                readStart = A.kmerOff[r];
            }
            mo = reinterpret_cast<const uint4 *>(A.mainOut)[r];
        }
        const bool live = inRange && mo.w != 0u && (cur[0].z & 31u) != 0u;
        const uint32_t mTax0 = mo.x, mTax1 = mo.y;
        QueryRec<16> Q;
        Q.decode_regs(cur, A.pool);
        if (!live) { Q.d = 0; Q.nInl = 0; Q.nMore = 0; Q.split = 0; Q.nseg = 0; }
        const bool split = Q.split != 0u;
        const int nEvMine = Q.d ? Q.d - A.kLow + 1 : 0;
        // ---- sweep 1: |T_k| of every query
        for (uint32_t w = 0; w < lvp_rows(nK); ++w) sLvN[wv][w][lane] = 0u;
        uint32_t incl = Q.nseg;
        incl = wave_incl_sum(incl);
        const uint32_t S1 = lane_value<63>(incl);
        if (S1 == 0u) continue;                                                // uniform: no live query
        sBase[wv][lane] = incl - Q.nseg;
        if (lane == 63) sBase[wv][64] = S1;
#pragma unroll
        for (int q = 0; q < INL; ++q) sSg[wv][q][lane] = Q.sg[q];
        sNInl[wv][lane] = Q.nInl;
        sPool[wv][lane] = Q.sg[INL - 1];
        LDS_WAVE_SYNC();
        for (uint32_t b0 = 0; b0 < S1; b0 += 64) {
            const uint32_t i = b0 + lane;
            if (i < S1) {
                uint32_t own = 0;
#pragma unroll
                for (int step = 32; step; step >>= 1) if (sBase[wv][own + step] <= i) own += step;
                const uint32_t idx = i - sBase[wv][own], nInl = sNInl[wv][own];
                const uint32_t sq = idx < nInl ? sSg[wv][idx][own] : A.pool[sPool[wv][own] + 1u + idx - nInl];
                const int la = A.kHigh - (int)(sq >> 27), lb = A.kHigh - (int)((sq >> 22) & 31u) + 1;
                atomicAdd(&sLvN[wv][la >> 1][own], lvp_unit(la));
                atomicSub(&sLvN[wv][lb >> 1][own], lvp_unit(lb));
            }
        }
        LDS_WAVE_SYNC();
        uint32_t bigLv = 0;                                                    // levels where the register taxa leave profile records
        lvp_running(&sLvN[wv][0][lane], 64, nK, [&](int lv, uint32_t n) { if (n > CNT_FIELDS) bigLv |= 1u << lv; });
        Q.tab = &sLvN[wv][0][lane]; Q.tabStride = 64;
        auto emitMask = [&](uint32_t sq) -> uint32_t {
            const uint32_t t = sq & SEG_TAX_MASK, m = seg_level_mask(sq, A.kHigh);
            return (t != mTax0 && t != mTax1) ? m : (GP ? 0u : (m & bigLv));
        };
        // ---- sweep 2: the records
        uint32_t mineSplit = 0;
        if (__ballot(split) != 0ull && split) {
#pragma unroll
            for (int q = 0; q < INL; ++q) if ((uint32_t)q < Q.nInl) mineSplit += (uint32_t)__popc(emitMask(Q.sg[q]));
            for (uint32_t q = 0; q < Q.nMore; ++q) mineSplit += (uint32_t)__popc(emitMask(Q.more[q]));
        }
        uint32_t emInl = 0;
#pragma unroll
        for (int q = 0; q < INL; ++q) if ((uint32_t)q < Q.nInl && emitMask(Q.sg[q]) != 0u) emInl |= 1u << q;
        const uint32_t nEmInl = (uint32_t)__popc(emInl);
        const uint32_t nFlat = split ? (mineSplit ? 1u : 0u) : nEmInl + Q.nMore;
        incl = nFlat;
        incl = wave_incl_sum(incl);
        const uint32_t S = lane_value<63>(incl);
        if (S == 0u) { LDS_WAVE_SYNC(); continue; }                            // uniform
        const bool head = inRange && (uint64_t)slot == readStart;
        const unsigned long long H = __ballot(head);
        const bool started = (H & upTo) != 0ull;
        LDS_WAVE_SYNC();                                                       // sweep 1's readers are done with sBase / sSg
        sBase[wv][lane] = incl - nFlat;
        if (lane == 63) sBase[wv][64] = S;
#pragma unroll
        for (int q = 0; q < INL; ++q) if ((emInl >> q) & 1u) sSg[wv][__popc(emInl & ((1u << q) - 1u))][lane] = Q.sg[q];
        sNInl[wv][lane] = nEmInl;
        sOrd[wv][0][lane] = cur[1].x; sOrd[wv][1][lane] = cur[1].y; sOrd[wv][2][lane] = cur[1].z; sOrd[wv][3][lane] = cur[1].w;
        sW2[wv][lane] = cur[0].z; sT0[wv][lane] = mTax0; sT1[wv][lane] = mTax1; sBig[wv][lane] = bigLv;
        sRow[wv][lane] = live ? mo.z + (started ? 0u : A.otherOff64[slot >> 6]) : 0u;
        sSplit[wv][lane] = mineSplit;
        LDS_WAVE_SYNC();
        uint32_t carry = 0;
        int prevOwnCarry = -1;
        for (uint32_t b0 = 0; b0 < S; b0 += 64) {
            const uint32_t i = b0 + lane;
            const bool act = i < S;
            uint32_t own = 0;
#pragma unroll
            for (int step = 32; step; step >>= 1) if (sBase[wv][own + step] <= i) own += step;
            if (!act) own = 63u;
            const uint32_t idx = i - sBase[wv][own];
            const uint32_t w2 = sW2[wv][own];
            const bool isSplit = (w2 & REC_SPLIT) != 0u;
            uint32_t sq = 0;
            const bool seg = act && !isSplit;
            if (seg) {
                const uint32_t nInl = sNInl[wv][own];
                sq = idx < nInl ? sSg[wv][idx][own] : A.pool[sPool[wv][own] + 1u + (INL - 1) + (idx - nInl) - (INL - 1)];
            }
            const uint32_t t = sq & SEG_TAX_MASK;
            const bool isMain = t == sT0[wv][own] || t == sT1[wv][own];
            const uint32_t m = seg ? seg_level_mask(sq, A.kHigh) : 0u;
            const uint32_t em = isMain ? (GP ? 0u : (m & sBig[wv][own])) : m;
            uint32_t c = (uint32_t)__popc(em);
            if (act && isSplit) c = sSplit[wv][own];
            const int prevOwn = lane_before((int)own, prevOwnCarry);
            const unsigned long long toOwn = own == 63u ? ~0ull : ((2ull << own) - 1ull);
            const unsigned long long toPrev = prevOwn < 0 ? 0ull : (prevOwn == 63 ? ~0ull : ((2ull << prevOwn) - 1ull));
            bool f = act && (H & toOwn & ~toPrev) != 0ull;
            uint32_t v = c;
            wave_seg_incl_sum(v, f);
            uint32_t w = sRow[wv][own] + v - c + (f ? 0u : carry);
            carry = lane_value<63>(v) + ((int)lane_value<63>((uint32_t)f) ? 0u : carry);
            prevOwnCarry = (int)lane_value<63>((uint32_t)own);
            if (act && c) {
                const uint32_t kind = isMain ? RK_PROFILE : kindOther;
                if (isSplit) sSplit[wv][own] = w;
                else if (c == 1u) {
                    const int lv = __ffs((int)em) - 1;
                    A.st[w] = make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | kind, (lvp_get(&sLvN[wv][0][own], 64, lv) << 16) | 1u);
                } else {                                                       // its levels in the query's flush order
                    const int nEv = (int)(w2 & 31u) - A.kLow + 1;
                    unsigned __int128 o = ((unsigned __int128)sOrd[wv][3][own] << 96) | ((unsigned __int128)sOrd[wv][2][own] << 64) |
                                          ((unsigned __int128)sOrd[wv][1][own] << 32) | sOrd[wv][0][own];
                    for (int ev = 0; ev < nEv; ++ev, o >>= 5) {
                        const int lv = (int)((uint32_t)o & 31u);
                        if ((em >> lv) & 1u) A.st[w++] = make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | kind, (lvp_get(&sLvN[wv][0][own], 64, lv) << 16) | 1u);
                    }
                }
            }
        }
        if (__ballot(split && mineSplit) != 0ull) {
            LDS_WAVE_SYNC();
            if (split && mineSplit) {                                          // a taxon may own several segments: event by event
                uint32_t w = sSplit[wv][lane];
                unsigned __int128 o = Q.order;
                auto putEvent = [&](uint32_t sq, int lv) {
                    const uint32_t t = sq & SEG_TAX_MASK;
                    const uint32_t kind = (t == mTax0 || t == mTax1) ? RK_PROFILE : kindOther;
                    A.st[w++] = make_uint2(t | ((uint32_t)lv << RK_LV_SHIFT) | kind, (Q.set_size(lv, A.kHigh) << 16) | 1u);
                };
                for (int ev = 0; ev < nEvMine; ++ev, o >>= 5) {
                    const int lv = (int)((uint32_t)o & 31u);
#pragma unroll
                    for (int q = 0; q < INL; ++q) if ((uint32_t)q < Q.nInl && ((emitMask(Q.sg[q]) >> lv) & 1u)) putEvent(Q.sg[q], lv);
                    for (uint32_t q = 0; q < Q.nMore; ++q) { const uint32_t sq = Q.more[q]; if ((emitMask(sq) >> lv) & 1u) putEvent(sq, lv); }
                }
            }
        }
        LDS_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------
// row_merge: one wavefront per staging row written by score_fast_kernel.  Sorts the row's records by
// (taxon, position) in LDS, sums each taxon's event scores IN THAT ORDER (= the read's flush order), and
// compacts the row in place to final {taxon, score} pairs, taxon ascending.  Every event / profile record
// also leaves as a 64-bit profile key {level:5 | |T|:13 | taxon:20 | hits:16} for the sort-reduce below.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void row_merge_kernel(const uint32_t *__restrict__ rowPos, uint32_t *__restrict__ rowLen,
                                                       const uint32_t *__restrict__ rowKey,
                                                       uint32_t nReads, uint2 *__restrict__ st, uint64_t *__restrict__ profKeys,
                                                       int kHigh, ProfLayout PL)
{
    __shared__ uint32_t sKey[RMAX];
    __shared__ uint16_t sIdx[RMAX];
    __shared__ uint2 sRec[RMAX];
    const int lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
        const uint32_t raw = rowLen[r];
        if (!(raw & ROW_MERGE)) continue;                              // uniform per block
        const uint32_t m = raw & ~ROW_MERGE;
        const uint32_t s0 = rowPos[r];
        uint32_t n2 = 2;
        while (n2 < m) n2 <<= 1;
        uint32_t keyAt = rowKey[r];
        for (uint32_t i0 = 0; i0 < n2; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint32_t key = 0xFFFFFFFFu;
            uint2 e = make_uint2(RK_FINAL, 0u);
            if (i < m) {
                e = st[s0 + i];
                sRec[i] = e;
                const uint32_t kind = e.x >> 30;
                if (kind != 2u) key = ((e.x & 0xFFFFFu) << 11) | (kind == 1u ? 0u : (i & 0x7FFu));   // final score first in its run
            }
            const uint32_t nk = (profKeys && i < m) ? record_keys(e) : 0u;          // (once: every event left as a profile key)
            uint32_t incl = nk;
            if (profKeys) incl = wave_incl_sum(incl);
            uint32_t kw = keyAt + incl - nk;
            if (profKeys) keyAt += lane_value<63>(incl);
            if (nk) {
                if ((e.x >> 30) == 3u) {
                    const uint32_t kF = (e.x >> 20) & 31u;
                    for (uint32_t k = kF; k <= (e.y & 31u); ++k)
                        profKeys[kw++] = profile_key_of((uint32_t)kHigh - k, seg_size(e.y, k == kF), e.x & 0xFFFFFu, 1u, PL);
                } else profKeys[kw] = profile_key(e, PL);
            }
            sKey[i] = key;
            sIdx[i] = (uint16_t)i;
        }
        LDS_WAVE_SYNC();
        for (uint32_t size = 2; size <= n2; size <<= 1)
            for (uint32_t stp = size >> 1; stp > 0; stp >>= 1) {
                for (uint32_t i = lane; i < n2; i += 64) {
                    const uint32_t j = i ^ stp;
                    if (j > i) {
                        const bool up = (i & size) == 0;
                        const uint32_t a = sKey[i], b = sKey[j];
                        if ((a > b) == up) {
                            sKey[i] = b; sKey[j] = a;
                            const uint16_t t = sIdx[i]; sIdx[i] = sIdx[j]; sIdx[j] = t;
                        }
                    }
                }
                LDS_WAVE_SYNC();
            }
        // runs of equal taxon; the lane that sees a run start replays the run
        uint32_t outBase = 0;
        for (uint32_t c0 = 0; c0 < m; c0 += 64) {
            const uint32_t i = c0 + lane;
            bool startsRun = false;
            float v = 0.0f;
            uint32_t tax = 0;
            if (i < m && sKey[i] != 0xFFFFFFFFu) {
                tax = sKey[i] >> 11;
                startsRun = (i == 0) || ((sKey[i - 1] >> 11) != tax);
                if (startsRun) {
                    bool any = false;
                    for (uint32_t q = i; q < m && sKey[q] != 0xFFFFFFFFu && (sKey[q] >> 11) == tax; ++q) {
                        const uint2 e = sRec[sIdx[q]];
                        if ((e.x >> 30) == 1u) { v = __uint_as_float(e.y); any = true; }
                        else if ((e.x >> 30) == 3u) {                  // a segment: its levels in ascending or descending order of k
                            const int kF = (int)((e.x >> 20) & 31u), kL = (int)(e.y & 31u);
                            const int step = (e.x & RK_SEG_DESC) ? -1 : 1;
                            for (int k = step > 0 ? kF : kL, c = kL - kF; c >= 0; --c, k += step)
                                v = __fadd_rn(v, event_score(k, seg_size(e.y, k == kF)));
                            any = true;
                        } else {
                            const float s = event_score(kHigh - (int)rk_level(e.x), e.y >> 16);
                            for (uint32_t j = 0; j < (e.y & 0xFFFFu); ++j) v = __fadd_rn(v, s);
                            any = true;
                        }
                    }
                    startsRun = any;
                }
            }
            const unsigned long long mk = __ballot(startsRun);
            if (startsRun) st[s0 + outBase + __popcll(mk & ((1ull << lane) - 1ull))] = make_uint2(tax, __float_as_uint(v));
            outBase += (uint32_t)__popcll(mk);
        }
        if (lane == 0) rowLen[r] = outBase;
        LDS_WAVE_SYNC();
    }
}

// The same for indices with at most BM_WORDS * 32 taxa, without sorting: a bitmap of the row's taxa gives every
// distinct taxon its rank (= its slot in the ascending output); only taxa with several records need the ordered
// replay, done by the lane that holds the taxon's first record.
// Two instantiations: rows of up to 512 records over up to 2048 taxa need only 8.5 KiB of LDS, so many rows are
// in flight per CU (the kernel is latency-bound: a handful of dependent global round trips per row); the rest
// takes the large instantiation.  A row is processed by exactly one of them (mLo < m <= RCAP).
static constexpr int BM_WORDS = 512;

// LONG: rows of ANY length over at most RCAP = BMW * 32 taxa -- the LDS arrays are per SLOT (distinct taxon of the row), and
// a row has no more slots than the index has taxa: the records stream through twice, whatever their number (a 10 kb read's
// row holds ten thousand chance matches; round 5 sent such reads to score_dense_kernel, a wavefront per read with two or
// three lanes busy).
template <int RCAP, int BMW, bool LONG = false>
__global__ __launch_bounds__(64) void row_merge_bitmap_kernel(const uint32_t *__restrict__ rowPos, uint32_t *__restrict__ rowLen,
                                                              const uint32_t *__restrict__ rowKey,
                                                              uint32_t nReads, uint2 *__restrict__ st, uint64_t *__restrict__ profKeys,
                                                              int kHigh, uint32_t nTaxa, uint32_t mLo, ProfLayout PL)
{
    __shared__ uint32_t bm[BMW], pre[BMW];
    __shared__ float val[RCAP];                                        // running score of a slot (= taxon of the row)
    __shared__ uint32_t slotTax[RCAP], claim[RCAP];
    __shared__ EventTables evT;
    event_tables_init(evT);
    const int lane = threadIdx.x;
    const uint32_t W = (nTaxa + 31u) >> 5;
    constexpr int HELD = 4;                                            // chunks of a row kept in registers between the passes
    // the row's whereabouts are fetched one row ahead: the kernel is a chain of dependent round trips per row
    uint32_t r = blockIdx.x;
    uint32_t nRaw = r < nReads ? rowLen[r] : 0u, nPos = r < nReads ? rowPos[r] : 0u, nKey = r < nReads ? rowKey[r] : 0u;
    for (; r < nReads; r += gridDim.x) {
        const uint32_t raw = nRaw, s0 = nPos, key0 = nKey;
        const uint32_t rNext = r + gridDim.x;
        if (rNext < nReads) { nRaw = rowLen[rNext]; nPos = rowPos[rNext]; nKey = rowKey[rNext]; }
        if (!(raw & ROW_MERGE)) continue;                              // uniform per block
        const uint32_t m = raw & ~ROW_MERGE;
        if (m <= mLo || (!LONG && m > (uint32_t)RCAP)) continue;
        uint2 held[HELD];                                              // all loads of the first chunks go out together
#pragma unroll
        for (int c = 0; c < HELD; ++c) {
            const uint32_t i = (uint32_t)c * 64u + lane;
            held[c] = i < m ? st[s0 + i] : make_uint2(RK_FINAL, 0u);
        }
        for (uint32_t w = lane; w < W; w += 64) bm[w] = 0u;
        for (uint32_t i = lane; i < (LONG ? min(m, (uint32_t)RCAP) : m); i += 64) { val[i] = 0.0f; claim[i] = 0xFFFFFFFFu; }
        LDS_WAVE_SYNC();
        // pass 1: the row's taxa as a bitmap; every event leaves as a profile key (a prefix sum places the keys of a chunk)
        uint32_t keyAt = key0;
        auto pass1 = [&](uint32_t i0, uint2 e) {
            const uint32_t i = i0 + lane;
            const uint32_t kind = e.x >> 30;
            const uint32_t t = e.x & 0xFFFFFu;
            if (i < m && kind != 2u) atomicOr(&bm[t >> 5], 1u << (t & 31u));
            const uint32_t nk = (profKeys && i < m) ? record_keys(e) : 0u;
            uint32_t incl = nk;
            if (profKeys) incl = wave_incl_sum(incl);                  // (uniform: a kernel argument)
            uint32_t kw = keyAt + incl - nk;
            if (profKeys) keyAt += lane_value<63>(incl);
            if (nk) {
                if (kind == 3u) {
                    const uint32_t kF = (e.x >> 20) & 31u;
                    for (uint32_t k = kF; k <= (e.y & 31u); ++k)
                        profKeys[kw++] = profile_key_of((uint32_t)kHigh - k, seg_size(e.y, k == kF), t, 1u, PL);
                } else profKeys[kw] = profile_key(e, PL);
            }
        };
#pragma unroll
        for (int c = 0; c < HELD; ++c) if ((uint32_t)c * 64u < m) pass1((uint32_t)c * 64u, held[c]);
        for (uint32_t i0 = HELD * 64u; i0 < m; i0 += 64) {
            const uint32_t i = i0 + lane;
            pass1(i0, i < m ? st[s0 + i] : make_uint2(RK_FINAL, 0u));
        }
        LDS_WAVE_SYNC();
        uint32_t carry = 0;                                            // exclusive popcount prefix over the bitmap words
        for (uint32_t w0 = 0; w0 < W; w0 += 64) {
            const uint32_t w = w0 + lane;
            const uint32_t pc = (w < W) ? (uint32_t)__popc(bm[w]) : 0u;
            uint32_t incl = pc;
            incl = wave_incl_sum(incl);
            if (w < W) pre[w] = carry + incl - pc;
            carry += lane_value<63>(incl);
        }
        const uint32_t nSlots = carry;
        LDS_WAVE_SYNC();
        // pass 2: 64 records at a time, in row order (= the read's flush order).  A record adds its hits to the running
        // score of its taxon's slot; two records of one chunk that share a slot take turns in lane order.
        auto pass2 = [&](uint32_t i0, uint2 e) {
            const uint32_t i = i0 + lane;
            bool pending = false;
            uint32_t slot = 0;
            float sc = 0.0f;
            if (i < m) {
                const uint32_t kind = e.x >> 30;
                if (kind != 2u) {
                    const uint32_t t = e.x & 0xFFFFFu;
                    slot = pre[t >> 5] + (uint32_t)__popc(bm[t >> 5] & ((1u << (t & 31u)) - 1u));
                    slotTax[slot] = t;                                 // every record of the slot writes the same value
                    if (kind == 1u) val[slot] = __uint_as_float(e.y);  // a register taxon's final score: all its events are in it
                    else { pending = true; if (kind == 0u) sc = event_score(evT, kHigh - (int)rk_level(e.x), e.y >> 16); }
                }
            }
            while (__ballot(pending) != 0ull) {
                if (pending) atomicMin(&claim[slot], (uint32_t)lane);
                LDS_WAVE_SYNC();
                if (pending && claim[slot] == (uint32_t)lane) {
                    float v = val[slot];
                    if ((e.x >> 30) == 3u) {                           // a segment: its levels in ascending or descending order of k
                        const int kF = (int)((e.x >> 20) & 31u), kL = (int)(e.y & 31u);
                        const int step = (e.x & RK_SEG_DESC) ? -1 : 1;
                        for (int k = step > 0 ? kF : kL, c = kL - kF; c >= 0; --c, k += step)
                            v = __fadd_rn(v, event_score(evT, k, seg_size(e.y, k == kF)));
                    } else
                        for (uint32_t j = 0; j < (e.y & 0xFFFFu); ++j) v = __fadd_rn(v, sc);
                    val[slot] = v;
                    claim[slot] = 0xFFFFFFFFu;
                    pending = false;
                }
                LDS_WAVE_SYNC();
            }
        };
#pragma unroll
        for (int c = 0; c < HELD; ++c) if ((uint32_t)c * 64u < m) pass2((uint32_t)c * 64u, held[c]);
        for (uint32_t i0 = HELD * 64u; i0 < m; i0 += 64) {
            const uint32_t i = i0 + lane;
            pass2(i0, i < m ? st[s0 + i] : make_uint2(0u, 0u));       // (L2-hot: read in pass 1 a moment ago)
        }
        LDS_WAVE_SYNC();
        for (uint32_t sl = lane; sl < nSlots; sl += 64) st[s0 + sl] = make_uint2(slotTax[sl], __float_as_uint(val[sl]));
        if (lane == 0) rowLen[r] = nSlots;                             // flag cleared: the other instantiation skips it
        LDS_WAVE_SYNC();
    }
}

// The sorted profile keys reduced in one streaming pass.  A distinct (level, |T|, taxon) has tens of thousands of
// consecutive keys, so almost every workgroup sees a single key: it sums the hits and one thread adds them to the
// tables.  A workgroup that holds a boundary lets every thread add the runs of its own eight keys.
static constexpr int PR_THREADS = 256, PR_ITEMS = 8;
__global__ __launch_bounds__(PR_THREADS) void profile_reduce_kernel(const uint64_t *__restrict__ sorted, uint32_t nKeys, uint32_t nTaxa,
                                                                     uint64_t *__restrict__ cntUnique, uint64_t *__restrict__ hiTab,
                                                                     uint64_t *__restrict__ midTab, uint64_t *__restrict__ loTab, ProfLayout PL)
{
    __shared__ unsigned long long sSum[PR_THREADS / 64];
    __shared__ uint64_t sEdge[2];
    const uint64_t mask = (1ull << PL.bits()) - 1ull;
    auto add = [&](uint64_t key, uint64_t c) {
        const uint32_t tax = (uint32_t)(key & ((1ull << PL.tb) - 1ull));
        if (c == 0 || tax == (1u << PL.tb) - 1u) return;               // unused slots (the key buffer starts as all ones)
        const uint32_t n = (uint32_t)((key >> PL.tb) & ((1ull << PL.nb) - 1ull));
        const uint32_t lv = (uint32_t)(key >> (PL.tb + PL.nb));
        const size_t cell = (size_t)lv * nTaxa + tax;
        if (n == 1) atomicAdd((unsigned long long *)&cntUnique[cell], (unsigned long long)c);
        fixed_add(hiTab, midTab, loTab, cell, c, n);
    };
    const uint32_t chunk = PR_THREADS * PR_ITEMS;
    for (uint64_t base = (uint64_t)blockIdx.x * chunk; base < nKeys; base += (uint64_t)gridDim.x * chunk) {
        const uint64_t i0 = base + threadIdx.x * PR_ITEMS;
        uint64_t v[PR_ITEMS];
        if (i0 + PR_ITEMS <= nKeys) {                                  // 64 bytes per thread as four 16-byte loads
#pragma unroll
            for (int j = 0; j < PR_ITEMS; j += 2) {
                const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(sorted + i0 + j);
                v[j] = w.x; v[j + 1] = w.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PR_ITEMS; ++j) v[j] = (i0 + j < nKeys) ? sorted[i0 + j] : ~0ull;
        }
        if (threadIdx.x == 0) sEdge[0] = (v[0] >> 16) & mask;
        const uint32_t lastIdx = (base + chunk <= nKeys) ? chunk - 1 : (uint32_t)(nKeys - 1 - base);
        if (threadIdx.x == lastIdx / PR_ITEMS) sEdge[1] = (v[lastIdx % PR_ITEMS] >> 16) & mask;
        __syncthreads();
        const bool uniform = sEdge[0] == sEdge[1];                    // sorted: equal ends mean one key throughout
        if (uniform) {
            unsigned long long sum = 0;
#pragma unroll
            for (int j = 0; j < PR_ITEMS; ++j) if (i0 + j < nKeys) sum += v[j] & 0xFFFFull;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
            if ((threadIdx.x & 63) == 0) sSum[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long total = 0;
                for (int w = 0; w < PR_THREADS / 64; ++w) total += sSum[w];
                add(sEdge[0], total);
            }
        } else {
            uint64_t cur = (v[0] >> 16) & mask, acc = 0;
#pragma unroll
            for (int j = 0; j < PR_ITEMS; ++j) {
                if (i0 + j >= nKeys) break;
                const uint64_t k = (v[j] >> 16) & mask;
                if (k != cur) { add(cur, acc); cur = k; acc = 0; }
                acc += v[j] & 0xFFFFull;
            }
            add(cur, acc);
        }
        __syncthreads();
    }
}

// The profile keys summed WITHOUT sorting them, when a (level, |T| <= NN, taxon) table of 32-bit counters fits the LDS
// of a workgroup: persistent workgroups stream the keys and count with LDS atomics (1.9e9 keys in ~3 ms).  Keys the table
// has no cell for -- a larger |T|, or so many hits that a 32-bit counter could wrap; a few per cent -- must not go to the global tables one by one
// (same-address atomics serialise: 200 ms): they collect in an LDS buffer that leaves with one cursor add per 1024+ keys,
// and only that short list is sorted and reduced.  At the end every workgroup adds its non-zero cells to the 64.64 tables.
static constexpr int PT_THREADS = 1024, PT_LEFT = 2048;
// Largest |T| the table counts per level, and where a level's cells start (in units of nTaxa): shallow levels have large
// taxon sets (chance matches of short prefixes), deep ones one or two taxa, so the cells are dealt out unevenly.
// One launch counts the keys of the levels lvLo .. lvHi - 1 (the others are skipped): with many levels (-k 25 7: 19) the
// cells of one workgroup do not go round, and several passes over the keys are far cheaper than sorting them all.
struct ProfTableLayout { uint8_t nn[MAX_LEVELS]; uint16_t first[MAX_LEVELS + 1]; int lvLo, lvHi; };
static ProfTableLayout prof_table_layout(int nK, int lvLo, int lvHi, uint32_t nTaxa, uint64_t budgetCells)
{
    ProfTableLayout L;
    memset(&L, 0, sizeof(L));
    L.lvLo = lvLo; L.lvHi = lvHi;
    const uint64_t perTaxon = nTaxa ? budgetCells / nTaxa : 0;         // cells every taxon can have over the window's levels
    const int nW = lvHi - lvLo;
    if (nW <= 0 || perTaxon < (uint64_t)nW) { L.lvHi = lvLo; return L; }   // not even |T| = 1 everywhere: no table
    uint64_t left = perTaxon - (uint64_t)nW;
    for (int lv = lvLo; lv < lvHi; ++lv) L.nn[lv] = 1;
    auto grow = [&](int lv, uint64_t upTo) { while (lv >= lvLo && lv < lvHi && L.nn[lv] < upTo && left > 0) { ++L.nn[lv]; --left; } };
    for (int lv = lvLo; lv < lvHi; ++lv) grow(lv, 2);                   // pairs (sibling taxa) at every level
    if (lvHi == nK) {                                                   // the window with the shallowest levels: large taxon sets
        grow(nK - 2, 4);
        grow(nK - 1, 24);                                               // the shallowest level takes what is left
        grow(nK - 2, 8);
    }
    for (uint64_t upTo = 3; upTo <= 8; ++upTo) for (int lv = lvHi - 1; lv >= lvLo; --lv) grow(lv, upTo);   // what is left: evenly, shallow first
    for (int lv = 0; lv < MAX_LEVELS; ++lv) L.first[lv + 1] = (uint16_t)(L.first[lv] + L.nn[lv]);
    return L;
}
__global__ __launch_bounds__(PT_THREADS) void profile_table_kernel(const uint64_t *__restrict__ keys, uint32_t nKeys, uint32_t nTaxa, int nK, ProfTableLayout TL,
                                                                   uint64_t *__restrict__ cntUnique, uint64_t *__restrict__ hiTab,
                                                                   uint64_t *__restrict__ midTab, uint64_t *__restrict__ loTab, ProfLayout PL,
                                                                   uint64_t *__restrict__ leftOut, unsigned long long *__restrict__ leftCursor)
{
    extern __shared__ uint32_t tab[];                                  // [level][|T| - 1][nTaxa], TL.nn[level] values of |T| per level
    __shared__ uint64_t sLeft[PT_LEFT];
    __shared__ uint32_t sLeftN;
    __shared__ unsigned long long sLeftBase;
    const uint32_t cells = (uint32_t)TL.first[MAX_LEVELS] * nTaxa;
    for (uint32_t i = threadIdx.x; i < cells; i += PT_THREADS) tab[i] = 0u;
    if (threadIdx.x == 0) sLeftN = 0;
    __syncthreads();
    const uint64_t mask = (1ull << PL.bits()) - 1ull;
    auto flush0 = [&]() {                                              // all threads; the buffer goes to the list
        const uint32_t n = min(sLeftN, (uint32_t)PT_LEFT);
        if (threadIdx.x == 0) sLeftBase = atomicAdd(leftCursor, (unsigned long long)n);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += PT_THREADS) leftOut[sLeftBase + i] = sLeft[i];   // (the list holds nKeys entries)
        __syncthreads();
        if (threadIdx.x == 0) sLeftN = 0;
        __syncthreads();
    };
    // A round = PT_KEYS keys per thread between two barriers (one key per round made the barrier the kernel's clock: 7 000
    // rounds per workgroup at 10 M reads).  Keys without a cell wait in sLeft; should a round bring more of them than the
    // buffer holds -- a few per cent of the keys have no cell, a round could in theory bring 8 192 -- the surplus goes to
    // the list one by one.
    constexpr int PT_KEYS = 8;
    const uint64_t step = (uint64_t)gridDim.x * PT_THREADS * PT_KEYS;
    // a workgroup sees at most `rounds * PT_THREADS * PT_KEYS` keys: with hits <= maxHits no 32-bit counter can wrap
    const uint64_t rounds = ((uint64_t)nKeys + step - 1) / step;       // the same for every workgroup: barriers inside
    const uint32_t maxHits = (uint32_t)std::min<uint64_t>(65535ull, 0xFFFFFFFFull / std::max<uint64_t>(1, rounds * PT_THREADS * PT_KEYS));
    auto &flush = flush0;
    for (uint64_t rd = 0; rd < rounds; ++rd) {
        const uint64_t i0 = rd * step + (uint64_t)blockIdx.x * PT_THREADS * PT_KEYS + threadIdx.x;
        uint64_t kk[PT_KEYS];
#pragma unroll
        for (int q = 0; q < PT_KEYS; ++q) { const uint64_t i = i0 + (uint64_t)q * PT_THREADS; kk[q] = i < nKeys ? keys[i] : 0ull; }   // (hits = 0: skipped)
#pragma unroll
        for (int q = 0; q < PT_KEYS; ++q) {
            const uint64_t key = kk[q];
            const uint64_t f = (key >> 16) & mask;
            const uint32_t hits = (uint32_t)(key & 0xFFFFull);
            const uint32_t tax = (uint32_t)(f & ((1ull << PL.tb) - 1ull));
            if (hits != 0u && tax != (1u << PL.tb) - 1u) {               // (else: unused slot)
                const uint32_t n = (uint32_t)((f >> PL.tb) & ((1ull << PL.nb) - 1ull));
                const uint32_t lv = (uint32_t)(f >> (PL.tb + PL.nb));
                if ((int)lv < TL.lvLo || (int)lv >= TL.lvHi) {}                // another pass counts this level
                else if (n >= 1u && n <= (uint32_t)TL.nn[lv] && hits <= maxHits) atomicAdd(&tab[((uint32_t)TL.first[lv] + (n - 1u)) * nTaxa + tax], hits);
                else {
                    const uint32_t at = atomicAdd(&sLeftN, 1u);
                    if (at < (uint32_t)PT_LEFT) sLeft[at] = key;
                    else leftOut[atomicAdd(leftCursor, 1ull)] = key;     // the buffer is full: straight to the list
                }
            }
        }
        __syncthreads();
        if (sLeftN > (uint32_t)(PT_LEFT / 2)) flush();                 // uniform: read after the barrier
    }
    flush();
    for (uint32_t i = threadIdx.x; i < cells; i += PT_THREADS) {
        const uint32_t c = tab[i];
        if (!c) continue;
        const uint32_t tax = i % nTaxa, row = i / nTaxa;
        uint32_t lv = 0;
        while (row >= (uint32_t)TL.first[lv + 1]) ++lv;
        const uint32_t n = row - (uint32_t)TL.first[lv] + 1u;
        const size_t cell = (size_t)lv * nTaxa + tax;
        if (n == 1u) atomicAdd((unsigned long long *)&cntUnique[cell], (unsigned long long)c);
        fixed_add(hiTab, midTab, loTab, cell, c, n);
    }
}

// final rows -> CSR in read order (rows are already {taxon, score}, taxon ascending)
__global__ __launch_bounds__(256) void row_copy_kernel(const uint32_t *__restrict__ rowPos, const uint32_t *__restrict__ rowLen,
                                const uint64_t *__restrict__ rowOff, uint32_t nReads, const uint2 *__restrict__ st,
                                uint32_t *__restrict__ outTax, float *__restrict__ outScore)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t waves = gridDim.x * 4u;
    for (uint32_t r = blockIdx.x * 4u + wv; r < nReads; r += waves) {
        const uint32_t s = rowPos[r];
        const uint32_t m = rowLen[r] & ~ROW_MERGE;
        const uint64_t o = rowOff[r];
        for (uint32_t i = lane; i < m; i += 64) { const uint2 e = st[s + i]; outTax[o + i] = e.x; outScore[o + i] = __uint_as_float(e.y); }
    }
}

__global__ void widen_kernel(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] & 0x7FFFFFFFu;   // bit 31 = ROW_MERGE
}

__global__ void invert_kernel(const uint32_t *__restrict__ plist, uint32_t n, uint32_t *__restrict__ slotOf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slotOf[plist[i]] = i;
}

// Slots from read ids (payload mode READ: kasa_batch_set_queries, -e, reads too long for the encoder's ranking): a stable
// sort of the sorted positions by read id lists every read's queries in sorted order; the inverse is the slot.
static int slots_from_reads(kasa_ctx *c)
{
    const uint64_t nQ = c->nQ;
    int rc;
    if ((rc = c->plist.reserve(nQ * 4 + 64)) || (rc = c->slotBuf.reserve(nQ * 4 + 64)) || (rc = c->qReadA.reserve(nQ * 4 + 64))) return rc;
    hipEvent_t a, b;
    if ((rc = timer_begin(c, c->timers[KASA_STAGE_REGROUP], &a, &b))) return rc;
    if (nQ) {
        unsigned bits = 1;
        while ((1ull << bits) < (uint64_t)std::max<int64_t>(c->nReads, 1)) ++bits;
        rocprim::counting_iterator<uint32_t> iota(0);
        size_t tmpBytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, tmpBytes, c->qRead, c->qReadA.as<uint32_t>(), iota, c->plist.as<uint32_t>(),
                                         (size_t)nQ, 0u, bits, c->stream));
        if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
        HIPCHK(rocprim::radix_sort_pairs(c->sortTmp.p, tmpBytes, c->qRead, c->qReadA.as<uint32_t>(), iota, c->plist.as<uint32_t>(),
                                         (size_t)nQ, 0u, bits, c->stream));
        invert_kernel<<<blocks_for(nQ, 256), 256, 0, c->stream>>>(c->plist.as<uint32_t>(), (uint32_t)nQ, c->slotBuf.as<uint32_t>());
        HIPCHK(hipGetLastError());
    }
    if ((rc = timer_end(c, c->timers[KASA_STAGE_REGROUP], a, b))) return rc;
    c->slotOf = c->slotBuf.as<uint32_t>();
    return KASA_OK;
}

template <int RW>
static int launch_group(kasa_ctx *c, const uint32_t *slotOf, uint32_t nTiles, uint32_t cap, uint32_t keyCap, int cov, unsigned long long *cursor, bool coop,
                        const uint32_t *tileList = nullptr, uint32_t nListed = 0)
{
    const uint64_t nQ = c->nQ;
    uint32_t *needCoop = reinterpret_cast<uint32_t *>(cursor + 3);        // (misc word [19]: free during the kernel)
    const uint32_t grid = tileList ? nListed : nTiles;                    // (a list: only the tiles group2_kernel left over)
#define KASA_GROUP_ARGS(KEY, META) c->keys<KEY>(), c->depth.as<uint8_t>(), c->rep.as<uint32_t>(), slotOf, (uint32_t)nQ, \
        c->tileNext.as<uint32_t>(), nTiles, c->ix->meta.as<META>(), c->ix->tax.as<uint32_t>(), (uint32_t)c->ix->n, \
        c->kHigh, c->kLow, c->recOut ? c->recOut : c->rec.as<uint32_t>(), c->pool.as<uint32_t>(), cap, cursor, cov, c->cntTotal.as<uint64_t>(), c->ix->nTaxa, \
        c->profKeys.as<uint64_t>(), keyCap, cursor + 2, prof_layout(c->ix->nTaxa, c->nK), c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>(), needCoop, tileList, c->recOut ? (uint32_t)RW : c->recCW
    if (c->ix->wide && RW == 16 && c->nK == 19)                       // the default -k 25 7 of a 128-bit index: loops over exactly 19 levels
        group_kernel<RW, key128, RW == 16 ? 19 : 0><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP_ARGS(key128, uint16_t));
    else if (c->ix->wide)
        group_kernel<RW, key128, 0><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP_ARGS(key128, uint16_t));
    else if (RW == 8 && c->nK == 6) {                                // the default -k 12 7: loops over exactly six levels
        if (coop) group_kernel<RW, uint64_t, RW == 8 ? 6 : 0, RW == 8><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP_ARGS(uint64_t, uint8_t));
        else group_kernel<RW, uint64_t, RW == 8 ? 6 : 0><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP_ARGS(uint64_t, uint8_t));
    } else if (coop && RW == 8)
        group_kernel<RW, uint64_t, 0, RW == 8><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP_ARGS(uint64_t, uint8_t));
    else
        group_kernel<RW, uint64_t, 0><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP_ARGS(uint64_t, uint8_t));
#undef KASA_GROUP_ARGS
    HIPCHK(hipGetLastError());
    return KASA_OK;
}

// group2_kernel over all tiles (narrow records); the tiles it lists are the caller's to give to group_kernel
static int launch_group2(kasa_ctx *c, const uint32_t *slotOf, uint32_t nTiles, uint32_t cap, uint32_t keyCap, int cov, unsigned long long *cursor, uint32_t *slowCount,
                         const uint32_t *tileIn = nullptr, uint32_t nIn = 0, uint32_t *slowListOut = nullptr)
{
    const uint64_t nQ = c->nQ;
#define KASA_GROUP2_ARGS(KEY, META) c->keys<KEY>(), c->depth.as<uint8_t>(), c->rep.as<uint32_t>(), slotOf, (uint32_t)nQ, \
        c->tileNext.as<uint32_t>(), nTiles, c->ix->meta.as<META>(), c->ix->tax.as<uint32_t>(), (uint32_t)c->ix->n, \
        c->kHigh, c->kLow, c->recOut ? c->recOut : c->rec.as<uint32_t>(), c->pool.as<uint32_t>(), cap, cursor, cov, c->ix->nTaxa, \
        c->profKeys.as<uint64_t>(), keyCap, cursor + 2, prof_layout(c->ix->nTaxa, c->nK), c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>(), \
        slowCount, slowListOut ? slowListOut : c->tileList.as<uint32_t>(), c->recOut ? 8u : c->recCW, tileIn
    static const size_t pad = getenv("KASA_G2_PADLDS") ? (size_t)atoi(getenv("KASA_G2_PADLDS")) : 0;   // (occupancy experiments: unused dynamic LDS)
    const uint32_t grid = tileIn ? nIn : nTiles;
    if (tileIn) group2_kernel<uint64_t, 6, 0, 4><<<grid, GTHREADS, 0, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));   // (the default level count, 64-bit keys: group_stage asks for nothing else)
    else
    if (c->ix->wide) group2_kernel<key128, 0><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(key128, uint16_t));
    else if (c->nK == 6) {                                                                                // the default -k 12 7
        // (timing variants of the kernel -- template parameter VAR: 1 straight-line key cutting, 2 a barrier before the record
        // stores, 4 no cursors -- are instantiated only in experiment builds: -DKASA_G2_VARIANTS; each is 100 KB of code object)
#ifdef KASA_G2_VARIANTS
        static const int var = getenv("KASA_G2_VAR") ? atoi(getenv("KASA_G2_VAR")) : 0;
        if (var == 1) group2_kernel<uint64_t, 6, 1><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));
        else if (var == 2) group2_kernel<uint64_t, 6, 2><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));
        else if (var == 3) group2_kernel<uint64_t, 6, 3><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));
        else if (var == 4) group2_kernel<uint64_t, 6, 4><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));
        else
#endif
        group2_kernel<uint64_t, 6><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));
    }
    else group2_kernel<uint64_t, 0><<<grid, GTHREADS, pad, c->stream>>>(KASA_GROUP2_ARGS(uint64_t, uint8_t));
#undef KASA_GROUP2_ARGS
    HIPCHK(hipGetLastError());
    return KASA_OK;
}


// profile_table_kernel for group keys: a key adds its hits to the cell of every level it spans -- range cells [first level][taxon]
// for the bulk (|T| = 1 down to the shallowest level: ONE add), per-level cells [level][|T| - 1][taxon] for the rest.  Levels
// without a cell in this workgroup's table leave as per-level keys of the classic layout (sorted and reduced afterwards).  With
// a crowded index that is most of them, so they leave without any per-key atomic: a round's keys are looked at twice -- first
// the cells are served and the leftovers counted, then (one running sum over the workgroup, ONE add to the list's cursor) every
// thread writes its leftovers to its own place.  Should the list be full: straight to the tables.
__global__ __launch_bounds__(PT_THREADS) void profile_group_table_kernel(const uint64_t *__restrict__ keys, uint32_t nKeys, uint32_t nTaxa, int nK, ProfTableLayout TL,
                                                                         uint64_t *__restrict__ cntUnique, uint64_t *__restrict__ hiTab,
                                                                         uint64_t *__restrict__ midTab, uint64_t *__restrict__ loTab, ProfLayout PL,
                                                                         uint64_t *__restrict__ leftOut, unsigned long long *__restrict__ leftCursor, unsigned long long leftCap,
                                                                         int rangeCells)
{
    extern __shared__ uint32_t tab[];                                  // [level][|T| - 1][nTaxa], TL.nn[level] values of |T| per level
    const uint32_t cells = (uint32_t)TL.first[MAX_LEVELS] * nTaxa;
    uint32_t *tabR = tab + cells;                                      // ... followed, when there is room (rangeCells), by [first level][nTaxa]
    const uint32_t rangeN = rangeCells ? (uint32_t)nK * nTaxa : 0u;
    __shared__ uint32_t sWave[PT_THREADS / 64];
    __shared__ unsigned long long sBase;
    for (uint32_t i = threadIdx.x; i < cells + rangeN; i += PT_THREADS) tab[i] = 0u;
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    auto direct = [&](uint32_t lv, uint32_t n, uint32_t tax, uint32_t hits) {
        const size_t cell = (size_t)lv * nTaxa + tax;
        if (n == 1u) atomicAdd((unsigned long long *)&cntUnique[cell], (unsigned long long)hits);
        fixed_add(hiTab, midTab, loTab, cell, hits, n);
    };
    constexpr int PT_KEYS = 8;
    const uint64_t step = (uint64_t)gridDim.x * PT_THREADS * PT_KEYS;
    const uint64_t rounds = ((uint64_t)nKeys + step - 1) / step;
    // a workgroup sees at most `rounds * PT_THREADS * PT_KEYS` keys: with hits <= maxHits no 32-bit counter can wrap
    const uint32_t maxHits = (uint32_t)std::min<uint64_t>(65535ull, 0xFFFFFFFFull / std::max<uint64_t>(1, rounds * PT_THREADS * PT_KEYS));
    for (uint64_t rd = 0; rd < rounds; ++rd) {
        const uint64_t i0 = rd * step + (uint64_t)blockIdx.x * PT_THREADS * PT_KEYS + threadIdx.x;
        uint64_t kk[PT_KEYS];
#pragma unroll
        for (int q = 0; q < PT_KEYS; ++q) { const uint64_t i = i0 + (uint64_t)q * PT_THREADS; kk[q] = i < nKeys ? keys[i] : 0ull; }   // (hits = 0: skipped)
        // what a key's level does: 0 nothing, 1 a cell, 2 left over
        auto levelsOf = [&](uint64_t key, auto cell, auto left) {
            const uint32_t hits = (uint32_t)(key & 0xFFFFull);
            if (hits == 0u) return;
            const uint32_t tax = (uint32_t)(key >> 16) & SEG_TAX_MASK, n = (uint32_t)(key >> 38) & 0x1FFFu;
            const uint32_t lvLo = (uint32_t)(key >> 51) & 31u, lvHi = lvLo + ((uint32_t)(key >> 56) & 31u);
            if (rangeN && n == 1u && lvHi + 1u == (uint32_t)nK && hits <= maxHits) { cell(&tabR[lvLo * nTaxa + tax], hits); return; }
            for (uint32_t lv = lvLo; lv <= lvHi; ++lv) {                     // (one pass over all levels: no level windows here)
                if ((int)lv >= TL.lvLo && (int)lv < TL.lvHi && n >= 1u && n <= (uint32_t)TL.nn[lv] && hits <= maxHits) cell(&tab[((uint32_t)TL.first[lv] + (n - 1u)) * nTaxa + tax], hits);
                else left(lv, n, tax, hits);
            }
        };
        uint32_t nLeft = 0;
        unsigned long long which = 0;                                        // the leftovers of this thread's keys: bit 8 q + level (up to 8 levels)
#pragma unroll
        for (int q = 0; q < PT_KEYS; ++q)
            levelsOf(kk[q], [&](uint32_t *c, uint32_t hits) { atomicAdd(c, hits); }, [&](uint32_t lv, uint32_t, uint32_t, uint32_t) { ++nLeft; which |= 1ull << (8 * q + (int)(lv & 7u)); });
        // the leftovers' places: running sum over the workgroup, one add to the list's cursor
        const uint32_t incl = wave_incl_sum(nLeft);
        if (lane == 63) sWave[wv] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < PT_THREADS / 64; ++w) { const uint32_t o = sWave[w]; if (w < wv) before += o; total += o; }
        if (total == 0u) { __syncthreads(); continue; }                      // (uniform)
        if (threadIdx.x == 0) sBase = atomicAdd(leftCursor, (unsigned long long)total);
        __syncthreads();
        unsigned long long at = sBase + before + incl - nLeft;
        if (nLeft && nK <= 8) {                                              // straight from the marks
#pragma unroll
            for (int q = 0; q < PT_KEYS; ++q) {
                uint32_t m = (uint32_t)(which >> (8 * q)) & 255u;
                const uint64_t key = kk[q];
                const uint32_t hits = (uint32_t)(key & 0xFFFFull), tax = (uint32_t)(key >> 16) & SEG_TAX_MASK, n = (uint32_t)(key >> 38) & 0x1FFFu;
                while (m) {
                    const uint32_t lv = (uint32_t)__ffs((int)m) - 1u;
                    m &= m - 1u;
                    if (at < leftCap) leftOut[at] = profile_key_of(lv, n, tax, hits, PL); else direct(lv, n, tax, hits);
                    ++at;
                }
            }
        } else if (nLeft) {
#pragma unroll
            for (int q = 0; q < PT_KEYS; ++q)
                levelsOf(kk[q], [&](uint32_t *, uint32_t) {}, [&](uint32_t lv, uint32_t n, uint32_t tax, uint32_t hits) {
                    if (at < leftCap) leftOut[at] = profile_key_of(lv, n, tax, hits, PL); else direct(lv, n, tax, hits);
                    ++at;
                });
        }
        __syncthreads();                                                     // (sWave, sBase are written again in the next round)
    }
    __syncthreads();
    if (rangeN)                                                        // the range cells folded into the tables: a running sum over the first levels
        for (uint32_t tax = threadIdx.x; tax < nTaxa; tax += PT_THREADS) {
            unsigned long long acc = 0;
            for (int lv = 0; lv < nK; ++lv) {
                acc += tabR[(uint32_t)lv * nTaxa + tax];
                if (!acc) continue;
                const size_t cell = (size_t)lv * nTaxa + tax;
                atomicAdd((unsigned long long *)&cntUnique[cell], acc);
                atomicAdd((unsigned long long *)&hiTab[cell], acc);        // (fixed_add with |T| = 1)
            }
        }
    for (uint32_t i = threadIdx.x; i < cells; i += PT_THREADS) {
        const uint32_t c = tab[i];
        if (!c) continue;
        const uint32_t tax = i % nTaxa, row = i / nTaxa;
        uint32_t lv = 0;
        while (row >= (uint32_t)TL.first[lv + 1]) ++lv;
        const uint32_t n = row - (uint32_t)TL.first[lv] + 1u;
        const size_t cell = (size_t)lv * nTaxa + tax;
        if (n == 1u) atomicAdd((unsigned long long *)&cntUnique[cell], (unsigned long long)c);
        fixed_add(hiTab, midTab, loTab, cell, c, n);
    }
}

// Group keys straight into the tables, whatever |T| is: a workgroup keeps, for a WINDOW of levels, the exact 64.64 sums of every
// (level, taxon) cell in LDS as the limbs the global tables hold -- A += low 32 bits of x, B += next 32 bits, C += x >> 64,
// U += hits iff |T| = 1, where x = hits * floor(2^64 / |T|) (fixed_add) -- 64-bit LDS counters that cannot wrap between two
// flushes (hits < 2^16, at most 2^22 keys between them), and adds them to the tables cell by cell.  No per-|T| cells, no
// leftover keys, no sort: with a crowded index (|T| in the tens and hundreds) nine keys in ten used to leave
// profile_group_table_kernel as per-level leftovers -- 3e9 of them sorted and reduced, 53 ms for 2 M reads.  floor(2^64 / n)
// comes from a table (rTab[n], n < 8192; [1] unused: |T| = 1 adds hits to C and U).  Compare.hpp:922-925.
static constexpr int PA_THREADS = 1024, PA_NMAX = 8192;
__global__ void profile_rtab_kernel(uint64_t *__restrict__ rTab)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (uint32_t)PA_NMAX) return;
    uint64_t R = n ? 0xFFFFFFFFFFFFFFFFull / n : 0ull;
    if (n && (n & (n - 1)) == 0) R += 1;                               // n divides 2^64 (n = 1: wraps to 0, never used)
    rTab[n] = R;
}
__global__ __launch_bounds__(PA_THREADS) void profile_group_accum_kernel(const uint64_t *__restrict__ keys, uint32_t nKeys, uint32_t nTaxa, int winLo, int winHi,
                                                                         const uint64_t *__restrict__ rTab, uint64_t *__restrict__ cntUnique, uint64_t *__restrict__ hiTab,
                                                                         uint64_t *__restrict__ midTab, uint64_t *__restrict__ loTab)
{
    extern __shared__ unsigned long long shAcc[];                      // [3][cells]: A, B, C | U << 32
    const uint32_t cells = (uint32_t)(winHi - winLo) * nTaxa;
    unsigned long long *sA = shAcc, *sB = shAcc + cells, *sCU = shAcc + 2 * (size_t)cells;
    auto flush = [&]() {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cells; i += PA_THREADS) {
            const unsigned long long a = sA[i], b = sB[i], cu = sCU[i];
            if ((a | b | cu) == 0ull) continue;
            const size_t cell = (size_t)winLo * nTaxa + i;             // (window rows are consecutive levels)
            if (a) atomicAdd((unsigned long long *)&loTab[cell], a);
            if (b) atomicAdd((unsigned long long *)&midTab[cell], b);
            if (cu & 0xFFFFFFFFull) atomicAdd((unsigned long long *)&hiTab[cell], cu & 0xFFFFFFFFull);
            if (cu >> 32) atomicAdd((unsigned long long *)&cntUnique[cell], cu >> 32);
            sA[i] = 0ull; sB[i] = 0ull; sCU[i] = 0ull;
        }
        __syncthreads();
    };
    for (uint32_t i = threadIdx.x; i < 3u * cells; i += PA_THREADS) shAcc[i] = 0ull;
    __syncthreads();
    const uint64_t step = (uint64_t)gridDim.x * PA_THREADS;
    uint32_t sinceFlush = 0;
    for (uint64_t i0 = (uint64_t)blockIdx.x * PA_THREADS; i0 < nKeys; i0 += step) {   // (uniform per workgroup)
        const uint64_t i = i0 + threadIdx.x;
        const uint64_t key = i < nKeys ? keys[i] : 0ull;
        const uint32_t hits = (uint32_t)(key & 0xFFFFull);
        if (hits) {
            const uint32_t tax = (uint32_t)(key >> 16) & SEG_TAX_MASK, n = (uint32_t)(key >> 38) & 0x1FFFu;
            int lo = (int)((key >> 51) & 31u), hi = lo + (int)((key >> 56) & 31u);
            lo = lo < winLo ? winLo : lo; hi = hi >= winHi ? winHi - 1 : hi;
            if (lo <= hi && hits >= 256u) {                                // (group keys carry at most 128 hits; whatever else: straight to the tables)
                for (int lv = lo; lv <= hi; ++lv) {
                    const size_t cell = (size_t)lv * nTaxa + tax;
                    if (n == 1u) atomicAdd((unsigned long long *)&cntUnique[cell], (unsigned long long)hits);
                    fixed_add(hiTab, midTab, loTab, cell, hits, n);
                }
            } else if (lo <= hi) {
                unsigned long long a = 0, b = 0, cu;
                if (n == 1u) cu = (unsigned long long)hits | ((unsigned long long)hits << 32);
                else {
                    const uint64_t R = rTab[n];
                    const uint64_t lo64 = (uint64_t)hits * R;
                    a = lo64 & 0xFFFFFFFFull; b = lo64 >> 32; cu = __umul64hi((uint64_t)hits, R);
                }
                for (int lv = lo; lv <= hi; ++lv) {
                    const uint32_t cell = (uint32_t)(lv - winLo) * nTaxa + tax;
                    if (a) atomicAdd(&sA[cell], a);
                    if (b) atomicAdd(&sB[cell], b);
                    if (cu) atomicAdd(&sCU[cell], cu);
                }
            }
        }
        sinceFlush += PA_THREADS;
        if (sinceFlush >= (1u << 22)) { flush(); sinceFlush = 0; }   // (C and U: at most 2^22 adds of less than 2^8 each between two flushes)
    }
    flush();
}

static int profile_from_keys(kasa_ctx *c, uint64_t nKeys, bool grouped)
{
    if (nKeys == 0) return KASA_OK;
    const uint32_t nTaxa = c->ix->nTaxa;
    const int nK = c->nK;
    int rc;
    hipStream_t ps = c->stream;
    static const char *timing = getenv("KASA_PROF_TIMING");               // diagnostics: the phases' times on stderr
    auto now = [&]() { if (timing) (void)hipStreamSynchronize(ps); return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    const ProfLayout PL = prof_layout(nTaxa, nK);
    // Which way: the per-|T| cells of round 4 serve sparse taxon sets a little better (C2: 8.0 against 8.4 ms -- one pass over
    // the keys, |T| = 1 keys one add for all their levels), the exact accumulation below whatever |T| is (crowded: 11 against
    // 53 ms).  A context on the cooperative group kernel, or whose last batch left a quarter of its keys over, takes the latter.
    // (test taps: 67108864 always the cells, 268435456 always the accumulation)
    const bool accum = grouped && !(c->debugFlags & 67108864) && (c->groupCoop || c->profLeftHint > nKeys / 4 || (c->debugFlags & 268435456));
    if (accum) {
        // exact accumulation of every cell in LDS, a window of levels per pass (24 bytes per cell)
        const uint64_t budget = 152u * 1024u;
        const int perPass = (int)std::min<uint64_t>((uint64_t)nK, budget / ((uint64_t)nTaxa * 24u));
        if (perPass >= 1 && (nK + perPass - 1) / perPass <= 3) {       // (more than three passes over the keys: the older way)
            if (!c->rTab.p) {
                if ((rc = c->rTab.reserve((size_t)PA_NMAX * 8))) return rc;
                profile_rtab_kernel<<<PA_NMAX / 256, 256, 0, ps>>>(c->rTab.as<uint64_t>());
                HIPCHK(hipGetLastError());
            }
            int nCu = 0;
            HIPCHK(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, c->device));
            for (int lo = 0; lo < nK; lo += perPass) {
                const int hi = std::min(nK, lo + perPass);
                const size_t shBytes = (size_t)(hi - lo) * nTaxa * 24;
                HIPCHK(hipFuncSetAttribute((const void *)profile_group_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shBytes));
                profile_group_accum_kernel<<<std::max(1, nCu), PA_THREADS, shBytes, ps>>>(c->profKeys.as<uint64_t>(), (uint32_t)nKeys, nTaxa, lo, hi,
                    c->rTab.as<uint64_t>(), c->cntUnique.as<uint64_t>(), c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>());
                HIPCHK(hipGetLastError());
            }
            if (timing) { const auto t1 = now(); fprintf(stderr, "kasa: profile keys %llu: exact LDS accumulation, %d level(s) per pass, %.1f ms\n", (unsigned long long)nKeys, perPass, std::chrono::duration<double>(t1 - t0).count() * 1e3); }
            return KASA_OK;
        }
    }
    // Keys without a cell in a workgroup's table leave as per-level keys for the sort + reduce: a few per cent of them with sparse
    // taxon sets, several per group key with a crowded index (|T| in the tens and hundreds).  The list is as long as the last
    // batch needed (what does not fit goes to the tables key by key -- slow, and the next batch gets the room).
    uint64_t leftCap = grouped ? std::max<uint64_t>(std::max<uint64_t>(nKeys / 4, 1u << 20), c->profLeftHint) : std::max<uint64_t>(nKeys, 1u << 20);
    leftCap = std::min<uint64_t>(leftCap, 0xFFFFFFF0ull);
    if ((rc = c->profSorted.reserve((size_t)leftCap * 8 + 64))) return rc;
    if (grouped && (rc = c->profSorted2.reserve((size_t)leftCap * 8 + 64))) return rc;
    uint64_t budgetCells = (160u * 1024u - PT_LEFT * 8u - 1024u) / 4u;
    // group keys: range cells [first level][taxon] for the keys with |T| = 1 (when a third of the table's room suffices)
    const bool rangeCells = grouped && (uint64_t)nK * nTaxa <= budgetCells / 3;
    if (rangeCells) budgetCells -= (uint64_t)nK * nTaxa;
    std::vector<ProfTableLayout> passes;
    if (nK <= 8 || grouped) passes.push_back(prof_table_layout(nK, 0, nK, nTaxa, budgetCells));
    else {
        passes.push_back(prof_table_layout(nK, nK - 4, nK, nTaxa, budgetCells));
        for (int hi = nK - 4; hi > 0; hi -= 8) passes.push_back(prof_table_layout(nK, std::max(0, hi - 8), hi, nTaxa, budgetCells));
    }
    bool tables = !(c->debugFlags & 16) || grouped;
    for (const auto &TL : passes) if (TL.lvHi <= TL.lvLo && !grouped) tables = false;      // a window without cells: everything is sorted
    uint64_t *sortIn = c->profKeys.as<uint64_t>(), *sortOut = c->profSorted.as<uint64_t>();
    uint64_t nSort = nKeys;
    if (tables) {
        int nCu = 0;
        HIPCHK(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, c->device));
        unsigned long long *leftCursor = c->misc.as<unsigned long long>() + 19;
        unsigned long long left = 0;
        HIPCHK(hipMemsetAsync(leftCursor, 0, 8, ps));
        for (const auto &TL : passes) {
            const size_t shBytes = ((size_t)TL.first[MAX_LEVELS] * nTaxa + (rangeCells ? (size_t)nK * nTaxa : 0)) * 4;
            HIPCHK(hipFuncSetAttribute((const void *)profile_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shBytes));
            if (grouped) {
                HIPCHK(hipFuncSetAttribute((const void *)profile_group_table_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shBytes));
                profile_group_table_kernel<<<std::max(1, nCu), PT_THREADS, shBytes, ps>>>(c->profKeys.as<uint64_t>(), (uint32_t)nKeys, nTaxa, nK, TL,
                    c->cntUnique.as<uint64_t>(), c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>(), PL,
                    c->profSorted.as<uint64_t>(), leftCursor, (unsigned long long)leftCap, rangeCells ? 1 : 0);
            } else
            profile_table_kernel<<<std::max(1, nCu), PT_THREADS, shBytes, ps>>>(c->profKeys.as<uint64_t>(), (uint32_t)nKeys, nTaxa, nK, TL,
                c->cntUnique.as<uint64_t>(), c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>(), PL,
                c->profSorted.as<uint64_t>(), leftCursor);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(&left, leftCursor, 8, hipMemcpyDeviceToHost, ps));
        HIPCHK(hipStreamSynchronize(ps));
        nSort = std::min<uint64_t>(left, leftCap); sortIn = c->profSorted.as<uint64_t>();
        sortOut = grouped ? c->profSorted2.as<uint64_t>() : c->profKeys.as<uint64_t>();   // (per-read keys: what is left is sorted back into the key buffer)
        if (grouped && left > leftCap) c->profLeftHint = left + left / 4;
    }
    const auto t1 = now();
    if (nSort > 0) {
        // keys only, by the bits above the 16-bit hit count (whole bytes: the bits beyond the key's fields are zero)
        const int sortBits = (int)((PL.bits() + 7u) / 8u) * 8;
        if ((rc = c->sortTmp.reserve(kasa_radix::scratch_bytes<uint64_t>(nSort)))) return rc;
        uint64_t *kRes = nullptr;
        HIPCHK(kasa_radix::sort_pairs<uint64_t>(sortIn, nullptr, sortOut, nullptr, (uint32_t)nSort, 16, sortBits, c->sortTmp.p, ps, &kRes, nullptr));
        profile_reduce_kernel<<<std::min<unsigned>(blocks_for(nSort, PR_THREADS * PR_ITEMS), 256u * 16u), PR_THREADS, 0, ps>>>(
            kRes, (uint32_t)nSort, nTaxa,
            c->cntUnique.as<uint64_t>(), c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>(), PL);
        HIPCHK(hipGetLastError());
    }
    if (timing) {
        const auto t2 = now();
        fprintf(stderr, "kasa: profile keys %llu: tables %.1f ms, %llu left over (list %llu), sort + reduce %.1f ms\n", (unsigned long long)nKeys,
                std::chrono::duration<double>(t1 - t0).count() * 1e3, (unsigned long long)nSort, (unsigned long long)leftCap, std::chrono::duration<double>(t2 - t1).count() * 1e3);
    }
    return KASA_OK;
}

// lookup_score = group (per sorted query: flush order of its events + taxon segments -> one record per query, in the query's
// read-major slot) followed by score (records replayed per read).  The two halves are separate entry points so that the
// records can travel: with a range-partitioned index the partition owner runs `group` on a slice
// of another rank's sorted queries and exports the records in sorted order, the read owner imports them into their slots
// and runs `score`.
static int group_stage(kasa_ctx *c, int coverage, bool exportSorted, uint32_t *recordsOut = nullptr)
{
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_group: batch not sorted");
    HIPCHK(hipSetDevice(c->ix->device));
    const uint64_t nQ = c->nQ;
    const int RW = c->recWords();
    int rc;
    c->haveScores = false; c->nnz = 0; c->grouped = false; c->poolUsed = 1; c->recSorted = exportSorted; c->recOut = exportSorted ? recordsOut : nullptr;
    if (nQ == 0) { c->grouped = true; return KASA_OK; }
    if (!exportSorted && !c->slotOf && (rc = slots_from_reads(c))) return rc;
    const uint32_t nTiles = (uint32_t)((nQ + TILE - 1) / TILE);
    // narrow records that go to random slots leave as whole 64-byte cells (group2_kernel, phase D) when the device has the room:
    // not for records exported in sorted order (a stream anyway), not under the older kernels' test taps, not with --coverage
    c->recCW = (uint32_t)RW;
    // MEASURED (round 6, C2): group2_kernel 54.8 -> 48.8 ms (its stores 25 -> 12.8 ms: 83 GB at the chip's write rate instead of 1.3e9
    // partial cells), but both record passes of the score stage then stream 83 GB instead of 41.6 and turn from issue-bound to
    // HBM-bound (score_main 16.9 -> 20.3 ms, score_other_flat 19.6 -> 23.9): the step is where it was, with 41.6 GB more resident.
    // So the cells are an OPTION (KASA_WIDE_CELLS=1, debug flag 134217728) until ONE pass reads the records; the default is 32-byte slots.
    static const bool cellsEnv = getenv("KASA_WIDE_CELLS") && atoi(getenv("KASA_WIDE_CELLS")) != 0;
    if (RW == 8 && !exportSorted && !c->noWideCells && !c->groupCoop && !coverage && (cellsEnv || (c->debugFlags & 134217728)) && !(c->debugFlags & (16777216 | 2048 | 131072 | 262144))) {
        if (c->rec.cap < nQ * (size_t)64 + 64) {
            size_t freeB = 0, totalB = 0;
            HIPCHK(hipMemGetInfo(&freeB, &totalB));
            if (freeB + c->rec.cap < nQ * (size_t)64 + 64 + ((size_t)24 << 30)) c->noWideCells = true;   // (24 GB are left to the later stages' buffers)
        }
        if (!c->noWideCells) c->recCW = 16u;
    }
    if (!c->recOut && (rc = (RW == 8 && c->recCW == 8u) ? reserve_placed(c->rec, nQ * (size_t)32 + 64, c->stream, &c->recPlacement)   // (32-byte slots: partial cells)
                                                          : c->rec.reserve(nQ * (size_t)c->recCW * 4 + 64))) return rc;
    unsigned long long *cursor = c->misc.as<unsigned long long>() + 16;   // 64-bit cursors: [16] pool, [17] staging, [18] profile keys
    hipEvent_t a, b;
    if (c->poolCap == 0) c->poolCap = std::max<uint64_t>(1u << 16, nQ);   // (a word per query: a first batch with crowded taxon lists -- 0.75 words per query on the bench data -- need not run group_kernel twice)
    if (c->keyCap == 0) c->keyCap = RW == 8 ? std::max<uint64_t>(1u << 20, nQ / 2 + nQ / 4) : 64;   // (group keys: narrow records only; at least 2^20: the buffer also takes the sorted leftovers of the table pass)
    uint64_t nKeys = 0;
    for (int attempt = 0;; ++attempt) {
        if (c->keyCap >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "the profile of this batch needs %llu keys (limit 2^32); split the batch", (unsigned long long)c->keyCap);
        if ((rc = c->pool.reserve(c->poolCap * 4)) || (rc = c->profKeys.reserve(c->keyCap * 8 + 64))) return rc;
        const unsigned long long one = 1, zero = 0;
        HIPCHK(hipMemcpyAsync(cursor, &one, 8, hipMemcpyHostToDevice, c->stream)); // offset 0 is never handed out
        HIPCHK(hipMemcpyAsync(cursor + 2, &zero, 8, hipMemcpyHostToDevice, c->stream));
        if ((rc = timer_begin(c, c->timers[KASA_STAGE_GROUP], &a, &b))) return rc;
        const uint32_t cap = (uint32_t)std::min<uint64_t>(c->poolCap, 0xFFFFFFF0ull);
        static const int g2tap = getenv("KASA_G2_TAP") ? atoi(getenv("KASA_G2_TAP")) : 0;   // (timing taps of group2_kernel: 32, 64, 128)
        static const int longSteps = getenv("KASA_LONG_STEPS") ? (atoi(getenv("KASA_LONG_STEPS")) & 127) : 0;   // (experiments: where a lane gives its walk to the wavefront)
        const int cov = (g2tap & (32 | 64 | 128 | 256)) | ((longSteps ? longSteps : (c->groupCoop ? (int)LONG_STEPS_CROWDED : 0)) << 9) | ((coverage && attempt == 0) ? 1 : 0) | ((c->debugFlags & 2048) ? 2 : 0) | ((c->debugFlags & 4096) ? 4 : 0) | ((c->debugFlags & 32768) ? 8 : 0) | ((c->debugFlags & 65536) ? 16 : 0);   // (test taps: no followers; no LDS span for 64-byte records)
        const uint32_t *slotOf = exportSorted ? nullptr : c->slotOf;
        hipEvent_t ka, kb;
        if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_GROUP], &ka, &kb))) return rc;
        HIPCHK(hipMemcpyAsync(cursor + 3, &zero, 8, hipMemcpyHostToDevice, c->stream));   // "a list too long for the lean kernel" | tiles group2_kernel lists << 32
        if (c->debugFlags & 262144) c->groupCoop = true;                   // (test tap 262144: the cooperative form from the first batch on)
        const bool coop = c->groupCoop && !(c->debugFlags & 131072);       // (test tap 131072: never the cooperative form)
        // narrow records: group2_kernel (leaders dense on the lanes) takes every ordinary tile and LISTS the others -- long taxon
        // lists, walks beyond its staged index span -- for group_kernel's cooperative form.  Not for a context whose batches are
        // mostly such tiles (groupCoop), not with --coverage, not under the test taps of the older kernel (16777216: never).
        const bool g2 = RW == 8 && !coop && !(cov & 1) && !(c->debugFlags & (16777216 | 2048 | 131072));
        unsigned long long used[4] = {0, 0, 0, 0};
        if (g2) {
            if ((rc = c->tileList.reserve((size_t)nTiles * 8 + 256))) return rc;   // (the first launch's list, then the second chance's)
            if ((rc = launch_group2(c, slotOf, nTiles, cap, (uint32_t)c->keyCap, cov, cursor, reinterpret_cast<uint32_t *>(cursor + 3) + 1))) return rc;
            HIPCHK(hipMemcpyAsync(used, cursor, 32, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            const uint32_t nListed = (uint32_t)(used[3] >> 32);
            c->lastSlowTiles = nListed; c->lastSlowTiles2 = nListed;
            if (nListed && getenv("KASA_DEBUG_WHY")) {
                std::vector<uint32_t> h(nListed);
                HIPCHK(hipMemcpy(h.data(), c->tileList.p, (size_t)nListed * 4, hipMemcpyDeviceToHost));
                uint32_t why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t x : h) why[(x >> 28) & 7u]++;
                fprintf(stderr, "[kasa] group2 left %u of %u tiles: long list %u, span %u, both %u, parked only %u, parked + long %u, parked + span %u, all %u\n",
                        nListed, nTiles, why[1], why[2], why[3], why[4], why[5], why[6], why[7]);
            }
            // The listed tiles (3 % at C2) are nearly all "parked too much": tiles that hold a heavy 7-letter group.  They get a second
            // chance with a park buffer four times as large (two workgroups per CU instead of three) before the cooperative kernel
            // -- 172 ns per tile there, 40 here -- takes what is left (long lists, walks beyond the span).
            const uint32_t *coopList = c->tileList.as<uint32_t>();
            uint32_t nCoop = nListed;
            static const bool g2Always = getenv("KASA_G2_ALWAYS") && atoi(getenv("KASA_G2_ALWAYS")) != 0;   // (experiment: group2_kernel + its second launch whatever the share of listed tiles)
            if (nListed && c->nK == 6 && !c->ix->wide && !(cov & (32 | 64 | 128 | 256)) && ((uint64_t)nListed * 4 <= nTiles || g2Always) && !getenv("KASA_NO_SECOND_CHANCE")) {
                uint32_t *count2 = c->misc.as<uint32_t>() + 78, *list2 = c->tileList.as<uint32_t>() + nTiles + 16;
                HIPCHK(hipMemsetAsync(count2, 0, 4, c->stream));
                if ((rc = launch_group2(c, slotOf, nTiles, cap, (uint32_t)c->keyCap, cov, cursor, count2, c->tileList.as<uint32_t>(), nListed, list2))) return rc;
                uint32_t n2 = 0;
                HIPCHK(hipMemcpyAsync(&n2, count2, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                coopList = list2; nCoop = n2;
                c->lastSlowTiles2 = n2;
                if (getenv("KASA_DEBUG_WHY")) fprintf(stderr, "[kasa] group2's second launch left %u of %u listed tiles to the cooperative kernel\n", n2, nListed);
            }
            if (nCoop && (rc = launch_group<8>(c, slotOf, nTiles, cap, (uint32_t)c->keyCap, cov, cursor, true, coopList, nCoop))) return rc;
            if ((uint64_t)nListed * 4 > nTiles && !(getenv("KASA_G2_ALWAYS") && atoi(getenv("KASA_G2_ALWAYS")) != 0)) c->groupCoop = true;       // crowded taxon lists: the context's further batches go to the older kernel directly
        } else {
            c->lastSlowTiles = 0; c->lastSlowTiles2 = 0;
            if ((rc = (RW == 8 ? launch_group<8>(c, slotOf, nTiles, cap, (uint32_t)c->keyCap, cov, cursor, coop) : launch_group<16>(c, slotOf, nTiles, cap, (uint32_t)c->keyCap, cov, cursor, false)))) return rc;
        }
        if ((rc = timer_end(c, c->kernels[KASA_KERNEL_GROUP], ka, kb))) return rc;
        if ((rc = timer_end(c, c->timers[KASA_STAGE_GROUP], a, b))) return rc;
        HIPCHK(hipMemcpyAsync(used, cursor, 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (g2 && (used[3] & 1ull)) return fail(KASA_E_LIMIT, "a query meets 255 or more taxa of a 128-bit index through narrow records; use the index's own k range");
        if (RW == 8 && !g2 && !coop && !(c->debugFlags & 131072) && ((used[3] & 1ull) || used[0] > 6ull * nQ)) {
            // a list of 255 or more segments (the lean kernel cannot count it), or long lists on average: whole wavefronts
            // take the long lists from here on -- this batch again if it must be, the context's further batches anyway
            c->groupCoop = true;
            if (used[3] & 1ull) { if (used[0] > c->poolCap) c->poolCap = used[0] + used[0] / 8 + 1024; continue; }
        }
        if (used[0] <= c->poolCap && used[2] <= c->keyCap) { c->poolUsed = (uint32_t)used[0]; nKeys = used[2]; break; }
        if (used[0] >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "the taxon lists of this batch need %llu pool entries (limit 2^32); split the batch", used[0]);
        if (used[0] > c->poolCap) c->poolCap = used[0] + used[0] / 8 + 1024;
        if (used[2] > c->keyCap) c->keyCap = used[2] + used[2] / 8 + 1024;
        if (attempt > 3) return fail(KASA_E_LIMIT, "taxon-list pool did not converge");
    }
    // the profile of the batch: the leaders' keys into the tables (the per-read score stage adds nothing to them)
    if ((rc = timer_begin(c, c->timers[KASA_STAGE_GROUP], &a, &b))) return rc;
    {
        hipEvent_t ka, kb;
        if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_PROFILE_TABLES], &ka, &kb))) return rc;
        if (RW == 8 && (rc = profile_from_keys(c, nKeys, true))) return rc;
        if ((rc = timer_end(c, c->kernels[KASA_KERNEL_PROFILE_TABLES], ka, kb))) return rc;
    }
    if ((rc = timer_end(c, c->timers[KASA_STAGE_GROUP], a, b))) return rc;
    c->lastKeys = nKeys;
    c->grouped = true;
    return KASA_OK;
}

template <int RW>
static int launch_flush(kasa_ctx *c, const uint32_t *list, uint32_t nList, const uint64_t *flushOff, uint32_t *out)
{
    const uint32_t nTiles = (uint32_t)((c->nQ + TILE - 1) / TILE);
    const unsigned blocks = std::min<uint32_t>(nList, 256u * 32u);
    if (c->ix->wide)
        flush_positions_kernel<RW, key128><<<blocks, 256, 0, c->stream>>>(list, nList, flushOff, c->kmerOff.as<uint64_t>(), c->rec.as<uint32_t>(),
            c->keys<key128>(), c->depth.as<uint8_t>(), (uint32_t)c->nQ, c->tileNext.as<uint32_t>(), nTiles, c->kHigh, c->kLow, out, c->recCW);
    else
        flush_positions_kernel<RW, uint64_t><<<blocks, 256, 0, c->stream>>>(list, nList, flushOff, c->kmerOff.as<uint64_t>(), c->rec.as<uint32_t>(),
            c->keys<uint64_t>(), c->depth.as<uint8_t>(), (uint32_t)c->nQ, c->tileNext.as<uint32_t>(), nTiles, c->kHigh, c->kLow, out, c->recCW);
    HIPCHK(hipGetLastError());
    return KASA_OK;
}

__global__ void list_counts_kernel(const uint32_t *__restrict__ list, uint32_t nList, const uint64_t *__restrict__ kmerOff, uint64_t *__restrict__ cnt,
                                   uint32_t *__restrict__ longest = nullptr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nList) {
        const uint32_t r = list[i];
        const uint64_t n = kmerOff[r + 1] - kmerOff[r];
        cnt[i] = n;
        if (longest) atomicMax(longest, (uint32_t)(n < 0xFFFFFFFFull ? n : 0xFFFFFFFFull));   // (the listed reads' most queries)
    }
    if (i == nList) cnt[i] = 0;
}

#include "kasa_replay.h"

static int score_stage(kasa_ctx *c, int wantPerRead)
{
    if (c->state < 3 || !c->grouped) return fail(KASA_E_STATE, "kasa_batch_score: no event records (call kasa_batch_group or kasa_batch_records_import)");
    if (c->recSorted) return fail(KASA_E_STATE, "kasa_batch_score: the records of this batch were exported in sorted order; import them on the read owner");
    HIPCHK(hipSetDevice(c->ix->device));
    const uint64_t nQ = c->nQ;
    const uint32_t nReads = (uint32_t)c->nReads;
    const uint32_t nTaxa = c->ix->nTaxa;
    const int nK = c->nK;
    const int RW = c->recWords();
    int rc;
    c->haveScores = false; c->nnz = 0;
    if (nQ == 0 || nReads == 0) {
        if (wantPerRead) {
            if ((rc = c->rowOff.reserve(((size_t)nReads + 1) * 8))) return rc;
            HIPCHK(hipMemsetAsync(c->rowOff.p, 0, ((size_t)nReads + 1) * 8, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            c->haveScores = true; c->csrPacked = true;
        }
        c->state = 4;
        return KASA_OK;
    }
    uint32_t *counters = c->misc.as<uint32_t>(); // [2] error flags, [3] fallback count, [5] second-pass count, [8..15] reasons; u64 [16] pool cursor, [17] staging cursor
    const bool gp = RW == 8;                     // narrow records: the profile is group_stage's, the kernels here leave no keys (they still add their zero)
    unsigned long long *stCursor = c->misc.as<unsigned long long>() + 17, *keyCursor = c->misc.as<unsigned long long>() + (gp ? 30 : 18);
    hipEvent_t a, b;

    if ((rc = c->rowPos.reserve((size_t)nReads * 4 + 64)) || (rc = c->rowLen.reserve((size_t)nReads * 4 + 64)) || (rc = c->rowKey.reserve((size_t)nReads * 4 + 64)) ||
        (rc = c->rowOff.reserve(((size_t)nReads + 1) * 8 + 64)) || (rc = c->fbList.reserve((size_t)nReads * 4 + 64)))
        return rc;
    if (c->stCap == 0) c->stCap = std::max<uint64_t>(1u << 16, (uint64_t)nReads * 8);
    const bool fast = nK <= 25 && nTaxa <= (1u << 20) && !c->forceSlowScore;   // staging records keep the taxon in 20 bits
    bool slowProfileDone = false;   // score_kernel adds to the profile tables itself: only once, whatever is rerun
    bool replayAddedProfile = false;
    c->lastOverflowReads = 0; c->lastThirdPassReads = 0;
    uint64_t staged = 0, nKeys = 0;
    if (!gp && c->keyCapScore == 0) c->keyCapScore = c->stCap;
    ScoreArgs A;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 4) return fail(KASA_E_LIMIT, "score staging did not converge");
        if (c->stCap >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "the score rows of this batch need %llu staging records (limit 2^32); split the batch", (unsigned long long)c->stCap);
        if (!gp && c->keyCapScore >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "the profile of this batch needs %llu keys (limit 2^32); split the batch", (unsigned long long)c->keyCapScore);
        if ((rc = c->st.reserve(c->stCap * 8)) || (!gp && (rc = c->profKeys.reserve(c->keyCapScore * 8 + 64)))) return rc;
        HIPCHK(hipMemsetAsync(counters + 2, 0, 8, c->stream));  // error flags, fallback count
        HIPCHK(hipMemsetAsync(counters + 8, 0, 32, c->stream));
        HIPCHK(hipMemsetAsync(stCursor, 0, 8, c->stream));             // staging cursor
        HIPCHK(hipMemsetAsync(keyCursor, 0, 8, c->stream));
        HIPCHK(hipMemsetAsync(c->rowLen.p, 0, (size_t)nReads * 4, c->stream));
        A.rec = c->rec.as<uint32_t>(); A.recCW = c->recCW; A.recQS = c->recCW == 8u ? 1u : 2u; A.kmerOff = c->kmerOff.as<uint64_t>();
        // long rows for long reads: the streaming row merge keeps a slot per taxon in LDS (up to 4096 taxa), the bitmap merge must be the one in use (test tap 4: the sorting merge)
        A.rowPerQuery = (nTaxa <= 4096u && !(c->debugFlags & 4) && !(getenv("KASA_NO_LONG_ROWS") && atoi(getenv("KASA_NO_LONG_ROWS")))) ? 8u : 0u;   // (KASA_NO_LONG_ROWS=1: round 5's limit, tests)
        A.pool = c->pool.as<uint32_t>(); A.nReads = nReads; A.kHigh = c->kHigh; A.kLow = c->kLow; A.nTaxa = nTaxa;
        A.scratch = nullptr; A.mainOut = nullptr; A.otherOff64 = nullptr; A.nQ = (uint32_t)nQ;
        A.cntUnique = c->cntUnique.as<uint64_t>(); A.cntAllHi = c->cntAllHi.as<uint64_t>(); A.cntAllMid = c->cntAllMid.as<uint64_t>(); A.cntAllLo = c->cntAllLo.as<uint64_t>();
        A.rowPos = c->rowPos.as<uint32_t>(); A.rowLen = c->rowLen.as<uint32_t>();
        A.st = c->st.as<uint2>();
        A.stCap = (uint32_t)c->stCap; A.stCursor = stCursor; A.errFlag = counters + 2;
        A.rowKey = c->rowKey.as<uint32_t>(); A.keyCap = gp ? 0xFFFFFFFFu : (uint32_t)c->keyCapScore; A.keyCursor = keyCursor;
        A.wantPerRead = wantPerRead ? 1 : 0;
        A.addProfile = gp ? 0 : (slowProfileDone ? 0 : 1);
        A.list = nullptr; A.nList = 0; A.flushPos = nullptr; A.flushOff = nullptr;
        A.fbList = c->fbList.as<uint32_t>(); A.fbCount = counters + 3; A.why = counters + 8;
        A.ovList = nullptr; A.ovCount = nullptr; A.workCursor = nullptr; A.forceHandOn = (c->debugFlags & 16384) ? 1 : 0;
        if ((rc = timer_begin(c, c->timers[KASA_STAGE_SCORE], &a, &b))) return rc;
        uint32_t nSlow = nReads;
        if (fast) {
            // persistent wavefronts: as many as are resident at once, each takes 64 reads at a time from a work counter
            int perCu = 0, nCu = 0;
            const bool fb8 = c->maxCnt <= 255u;                              // 8-bit counter fields: twice the wavefronts per CU
            const bool lv19 = RW == 16 && nK <= 19;                          // (the default -k 25 7 has 19 levels)
            typedef void (*MainKernel)(ScoreArgs);
            const bool lv6 = RW == 8 && nK <= 6;                             // (the default -k 12 7 has 6 levels: smaller LDS tables, one more wavefront per SIMD)
            const MainKernel kern = (RW == 8 && fb8 && wantPerRead && (c->debugFlags & 1024)) ? score_main_kernel<8, true, 8, 8, true> :
                                    (lv6 && fb8) ? (wantPerRead ? score_main_kernel<8, true, 8, 6> : score_main_kernel<8, false, 8, 6>) :
                                    RW == 8 ? (fb8 ? (wantPerRead ? score_main_kernel<8, true, 8> : score_main_kernel<8, false, 8>)
                                                   : (wantPerRead ? score_main_kernel<8, true> : score_main_kernel<8, false>))
                                  : lv19 ? (fb8 ? (wantPerRead ? score_main_kernel<16, true, 8, 19> : score_main_kernel<16, false, 8, 19>)
                                                : (wantPerRead ? score_main_kernel<16, true, 16, 19> : score_main_kernel<16, false, 16, 19>))
                                         : (fb8 ? (wantPerRead ? score_main_kernel<16, true, 8> : score_main_kernel<16, false, 8>)
                                                : (wantPerRead ? score_main_kernel<16, true> : score_main_kernel<16, false>));
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, (const void *)kern, 64, 0));
            HIPCHK(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, c->device));
            const uint32_t fblocks = std::min<uint32_t>((nReads + 63) / 64, (uint32_t)std::max(1, perCu) * (uint32_t)std::max(1, nCu));
            if ((rc = c->fastScratch.reserve((size_t)nReads * 16 + 64))) return rc;
            A.mainOut = c->fastScratch.as<uint32_t>();
            if ((rc = c->plist.reserve((nQ / 64 + 2) * 4 + 64))) return rc;      // (free once the slots are known)
            A.otherOff64 = c->plist.as<uint32_t>();
            A.workCursor = counters + 6;
            HIPCHK(hipMemsetAsync(counters + 6, 0, 4, c->stream));
            const unsigned oblocks = std::min<unsigned>(blocks_for(nQ, 256), 256u * 64u);
            hipEvent_t ka, kb;
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_SCORE_MAIN], &ka, &kb))) return rc;
            kern<<<fblocks, 64, 0, c->stream>>>(A);
            HIPCHK(hipGetLastError());
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_SCORE_MAIN], ka, kb))) return rc;
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_SCORE_OTHER], &ka, &kb))) return rc;
            const bool flat = RW == 8 && !(c->debugFlags & 32);              // debug flag 32: the per-lane kernels
            if (RW == 16 && !(c->debugFlags & 32)) {
                const unsigned fblocks16 = std::min<unsigned>(blocks_for(nQ, 128), 256u * 128u);
                if (nK <= 19) {
                    if (wantPerRead) score_other_flat16_kernel<true, 19><<<fblocks16, 128, 0, c->stream>>>(A);
                    else score_other_flat16_kernel<false, 19><<<fblocks16, 128, 0, c->stream>>>(A);
                } else if (wantPerRead) score_other_flat16_kernel<true><<<fblocks16, 128, 0, c->stream>>>(A);
                else score_other_flat16_kernel<false><<<fblocks16, 128, 0, c->stream>>>(A);
            } else
            if (wantPerRead) { if (flat) score_other_flat_kernel<true><<<oblocks, 256, 0, c->stream>>>(A); else if (RW == 8) score_other_kernel<8, true><<<oblocks, 256, 0, c->stream>>>(A); else score_other_kernel<16, true><<<oblocks, 256, 0, c->stream>>>(A); }
            else { if (flat) score_other_flat_kernel<false><<<oblocks, 256, 0, c->stream>>>(A); else if (RW == 8) score_other_kernel<8, false><<<oblocks, 256, 0, c->stream>>>(A); else score_other_kernel<16, false><<<oblocks, 256, 0, c->stream>>>(A); }
            HIPCHK(hipGetLastError());
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_SCORE_OTHER], ka, kb))) return rc;
            uint32_t h3 = 0; unsigned long long want[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(&h3, counters + 3, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(want, stCursor, gp ? 8 : 16, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            nSlow = h3;
            if (want[0] > c->stCap || (!gp && want[1] > c->keyCapScore)) {   // the fast kernels have no side effects: grow and rerun
                if ((rc = timer_end(c, c->timers[KASA_STAGE_SCORE], a, b))) return rc;
                if (want[0] > c->stCap) c->stCap = want[0] + want[0] / 8 + (uint64_t)nSlow * 64 + 1024;
                if (!gp && want[1] > c->keyCapScore) c->keyCapScore = want[1] + want[1] / 8 + 1024;
                continue;
            }
            nKeys = gp ? 0 : want[1];
            A.list = c->fbList.as<uint32_t>(); A.nList = nSlow;
        }
        c->lastDenseReads = 0;
        if (fast && nSlow > 0 && gp && wantPerRead && nTaxa * 5u * SD_WAVES <= 160u * 1024u - 1024u && !(c->debugFlags & 33554432)) {   // (the rows of a workgroup's reads fit LDS; test tap 33554432: never)
            // reads the fast kernels handed over (long rows, as a rule): those that keep the order rule need no pending window
            hipEvent_t da, db;
            if ((rc = c->fbList2.reserve((size_t)nSlow * 4 + 64))) return rc;
            uint32_t *handCount = counters + 70;
            HIPCHK(hipMemsetAsync(handCount, 0, 4, c->stream));
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_SCORE_DENSE], &da, &db))) return rc;
            static const int denseTap = getenv("KASA_DENSE_TAP") ? atoi(getenv("KASA_DENSE_TAP")) : 0;
            const int keepHandOn = A.forceHandOn;
            A.forceHandOn = denseTap;
            const size_t rowLds = ((size_t)nTaxa * 4 + (((size_t)nTaxa + 3) / 4) * 4) * SD_WAVES;
            const uint32_t dblocks = std::min<uint32_t>((nSlow + SD_WAVES - 1) / SD_WAVES, 256u * 8u);
            if (nK <= 6) {
                HIPCHK(hipFuncSetAttribute((const void *)score_dense_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                score_dense_kernel<6><<<dblocks, 64 * SD_WAVES, rowLds, c->stream>>>(A, c->fbList2.as<uint32_t>(), handCount);
            } else {
                HIPCHK(hipFuncSetAttribute((const void *)score_dense_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                score_dense_kernel<8><<<dblocks, 64 * SD_WAVES, rowLds, c->stream>>>(A, c->fbList2.as<uint32_t>(), handCount);
            }
            HIPCHK(hipGetLastError());
            A.forceHandOn = keepHandOn;
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_SCORE_DENSE], da, db))) return rc;
            uint32_t handed = 0;
            HIPCHK(hipMemcpyAsync(&handed, handCount, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            c->lastDenseReads = nSlow - handed;
            nSlow = handed;
            A.list = c->fbList2.as<uint32_t>(); A.nList = nSlow;
        }
        c->lastReplayReads = 0; c->lastReplayEvents = 0;
        if (nSlow > 0 && (wantPerRead || !gp) && !(c->debugFlags & 536870912)) {   // (test tap 536870912: never)
            // very long reads: events sorted by (read, taxon, flush position, level), one float chain per (read, taxon) (kasa_replay.h)
            const uint32_t minEnv = getenv("KASA_ESR_MIN_KMERS") ? (uint32_t)strtoul(getenv("KASA_ESR_MIN_KMERS"), nullptr, 10) : 0u;   // (tests: short reads through it)
            const uint32_t minK = (c->debugFlags & 1073741824) ? 1u : (minEnv ? minEnv : ESR_MIN_KMERS);   // (test tap 1073741824: every read of the general kernel's list)
            hipEvent_t ea, eb;
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_SCORE_REPLAY], &ea, &eb))) return rc;
            const uint32_t *rest = nullptr; uint32_t nRest = 0;
            if ((rc = esr_stage(c, A, nSlow, minK, &rest, &nRest, counters))) return rc;
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_SCORE_REPLAY], ea, eb))) return rc;
            if (c->lastReplayReads) { A.list = rest; A.nList = nRest; nSlow = nRest; if (!gp) replayAddedProfile = true; }
        }
        hipEvent_t ga = nullptr, gb = nullptr;
        if (nSlow > 0 && (rc = timer_begin(c, c->kernels[KASA_KERNEL_SCORE_GENERAL], &ga, &gb))) return rc;
        if (nSlow > 0) {
            // flush positions of the listed reads' queries (the records hold only the order inside a query)
            const uint64_t *flushOff = c->kmerOff.as<uint64_t>();
            uint64_t nFq = nQ;
            if (A.list) {
                if ((rc = c->flushOff.reserve(((size_t)nSlow + 1) * 8 + 64))) return rc;
                list_counts_kernel<<<blocks_for((uint64_t)nSlow + 1, 256), 256, 0, c->stream>>>(A.list, nSlow, c->kmerOff.as<uint64_t>(), c->flushOff.as<uint64_t>());
                size_t tmpBytes = 0;
                HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, c->flushOff.as<uint64_t>(), c->flushOff.as<uint64_t>(), (uint64_t)0, (size_t)nSlow + 1, rocprim::plus<uint64_t>(), c->stream));
                if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
                HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, c->flushOff.as<uint64_t>(), c->flushOff.as<uint64_t>(), (uint64_t)0, (size_t)nSlow + 1, rocprim::plus<uint64_t>(), c->stream));
                HIPCHK(hipMemcpyAsync(&nFq, c->flushOff.as<uint64_t>() + nSlow, 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                flushOff = c->flushOff.as<uint64_t>();
            }
            if ((rc = c->flushPos.reserve(nFq * (size_t)nK * 4 + 64))) return rc;
            if ((rc = (RW == 8 ? launch_flush<8>(c, A.list, nSlow, flushOff, c->flushPos.as<uint32_t>()) : launch_flush<16>(c, A.list, nSlow, flushOff, c->flushPos.as<uint32_t>())))) return rc;
            A.flushPos = c->flushPos.as<uint32_t>(); A.flushOff = flushOff;
            const uint64_t rowBytes = (uint64_t)nTaxa * 4;
            const uint32_t maxBlocks = (uint32_t)std::max<uint64_t>(1024, std::min<uint64_t>(256u * 32u, (8ull << 30) / rowBytes));
            const uint32_t blocks = std::min<uint32_t>(nSlow, maxBlocks);
            if ((rc = c->scratch.reserve((size_t)blocks * rowBytes)) || (rc = c->ovList.reserve((size_t)nSlow * 4 + 64))) return rc;
            HIPCHK(hipMemsetAsync(c->scratch.p, 0, (size_t)blocks * rowBytes, c->stream));
            HIPCHK(hipMemsetAsync(counters + 5, 0, 4, c->stream));
            A.scratch = c->scratch.as<float>();
            A.ovList = c->ovList.as<uint32_t>(); A.ovCount = counters + 5;
            const bool dense = nTaxa <= (uint32_t)DENSE_TAXA && !(c->debugFlags & 8192);   // (test tap 8192: the lane-owns-its-cells form)
            const size_t rowLds = dense ? (size_t)nTaxa * 4 : 0;
            if (dense) {
                HIPCHK(hipFuncSetAttribute((const void *)score_kernel<PCAP_SMALL, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                HIPCHK(hipFuncSetAttribute((const void *)score_kernel<PCAP_SMALL, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                HIPCHK(hipFuncSetAttribute((const void *)score_kernel<PCAP, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                HIPCHK(hipFuncSetAttribute((const void *)score_kernel<PCAP, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                if (RW == 8) score_kernel<PCAP_SMALL, 8, true><<<blocks, 64, rowLds, c->stream>>>(A);
                else score_kernel<PCAP_SMALL, 16, true><<<blocks, 64, rowLds, c->stream>>>(A);
            } else
            if (RW == 8) score_kernel<PCAP_SMALL, 8><<<blocks, 64, 0, c->stream>>>(A);      // leaves every score row zeroed again
            else score_kernel<PCAP_SMALL, 16><<<blocks, 64, 0, c->stream>>>(A);
            HIPCHK(hipGetLastError());
            uint32_t nOver = 0;
            HIPCHK(hipMemcpyAsync(&nOver, counters + 5, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            // Later passes over a list of reads: the flush positions are addressed through the list, so its offsets are made anew.
            // which = 2: the window of PCAP groups in LDS; narrow records hand a read whose window overflows even that to
            // which = 3: the window in device memory (GWIN), as long as the longest read has queries times levels.
            auto laterPass = [&](const uint32_t *list, uint32_t n, int which, uint32_t *ovOut, uint32_t *ovCnt) -> int {
                int rc2;
                if ((rc2 = c->flushOff2.reserve(((size_t)n + 1) * 8 + 64))) return rc2;
                uint32_t *longest = counters + 4;
                HIPCHK(hipMemsetAsync(longest, 0, 4, c->stream));
                list_counts_kernel<<<blocks_for((uint64_t)n + 1, 256), 256, 0, c->stream>>>(list, n, c->kmerOff.as<uint64_t>(), c->flushOff2.as<uint64_t>(), longest);
                size_t tmpBytes = 0;
                HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, c->flushOff2.as<uint64_t>(), c->flushOff2.as<uint64_t>(), (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), c->stream));
                if ((rc2 = c->sortTmp.reserve(tmpBytes))) return rc2;
                HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, c->flushOff2.as<uint64_t>(), c->flushOff2.as<uint64_t>(), (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), c->stream));
                uint64_t nFq2 = 0; uint32_t hLongest = 0;
                HIPCHK(hipMemcpyAsync(&nFq2, c->flushOff2.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipMemcpyAsync(&hLongest, longest, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if ((rc2 = c->flushPos2.reserve(nFq2 * (size_t)nK * 4 + 64))) return rc2;
                if ((rc2 = (RW == 8 ? launch_flush<8>(c, list, n, c->flushOff2.as<uint64_t>(), c->flushPos2.as<uint32_t>()) : launch_flush<16>(c, list, n, c->flushOff2.as<uint64_t>(), c->flushPos2.as<uint32_t>())))) return rc2;
                A.list = list; A.nList = n;
                A.flushPos = c->flushPos2.as<uint32_t>(); A.flushOff = c->flushOff2.as<uint64_t>();
                A.ovList = ovOut; A.ovCount = ovCnt;
                A.forceHandOn = (ovOut && (c->debugFlags & 8388608)) ? 1 : 0;   // (test tap 8388608: the second pass hands every read to the third)
                if (which == 3) {
                    // (per block four arrays as long as the batch's longest read has queries times levels: no read can overflow them)
                    const uint64_t cap = std::max<uint64_t>(64, (uint64_t)hLongest * (uint64_t)nK + 64);
                    const uint32_t b3 = std::min<uint32_t>(n, (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(256, (2ull << 30) / (cap * 16))));
                    if (cap > 0xFFFFFFFFull) return fail(KASA_E_LIMIT, "a read of %llu k-mers keeps more groups pending than this build can hold", (unsigned long long)hLongest);
                    if ((rc2 = c->gwin.reserve((size_t)b3 * cap * 16 + 64))) return rc2;
                    A.gwin = c->gwin.as<uint32_t>(); A.gwinCap = (uint32_t)cap;
                    if (dense) {
                        HIPCHK(hipFuncSetAttribute((const void *)score_kernel<1, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowLds));
                        score_kernel<1, 8, true, true><<<b3, 64, rowLds, c->stream>>>(A);
                    } else score_kernel<1, 8, false, true><<<b3, 64, 0, c->stream>>>(A);
                } else {
                    const uint32_t b2 = std::min<uint32_t>(n, std::min<uint32_t>(blocks, 256u * 16u));
                    if (dense) {
                        if (RW == 8) score_kernel<PCAP, 8, true><<<b2, 64, rowLds, c->stream>>>(A);
                        else score_kernel<PCAP, 16, true><<<b2, 64, rowLds, c->stream>>>(A);
                    } else
                    if (RW == 8) score_kernel<PCAP, 8><<<b2, 64, 0, c->stream>>>(A);
                    else score_kernel<PCAP, 16><<<b2, 64, 0, c->stream>>>(A);
                }
                HIPCHK(hipGetLastError());
                return KASA_OK;
            };
            if (nOver > 0) {
                // second pass.  64-byte records add to the profile as they go (nothing to take back): their window must hold --
                // KASA_E_LIMIT otherwise.  Narrow records leave no trace of a read they hand on (the profile is group_stage's).
                uint32_t *ov2 = nullptr, *ov2Cnt = nullptr;
                if (RW == 8 && gp) {
                    if ((rc = c->ovList2.reserve((size_t)nOver * 4 + 64))) return rc;
                    ov2 = c->ovList2.as<uint32_t>(); ov2Cnt = counters + 7;
                    HIPCHK(hipMemsetAsync(ov2Cnt, 0, 4, c->stream));
                }
                if ((rc = laterPass(c->ovList.as<uint32_t>(), nOver, 2, ov2, ov2Cnt))) return rc;
                if (ov2) {
                    uint32_t nOver2 = 0;
                    HIPCHK(hipMemcpyAsync(&nOver2, ov2Cnt, 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(hipStreamSynchronize(c->stream));
                    if (nOver2 > 0 && (rc = laterPass(ov2, nOver2, 3, nullptr, nullptr))) return rc;
                    c->lastThirdPassReads = std::max(c->lastThirdPassReads, nOver2);
                }
            }
            c->lastOverflowReads = std::max(c->lastOverflowReads, nOver);   // over the staging retries of this batch
            slowProfileDone = true;
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_SCORE_GENERAL], ga, gb))) return rc;
        }
        if (replayAddedProfile) slowProfileDone = true;              // (64-byte records: the replay added its reads' events to the tables -- once, whatever is rerun)
        c->lastSlowReads = nSlow;
        if ((rc = timer_end(c, c->timers[KASA_STAGE_SCORE], a, b))) return rc;
        uint32_t err = 0; unsigned long long want = 0;
        HIPCHK(hipMemcpyAsync(&err, counters + 2, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&want, stCursor, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (err & 2u) return fail(KASA_E_LIMIT, "a read keeps more than %d distinct groups pending; this build cannot order it", PCAP);
        if (want <= c->stCap) { staged = want; break; }
        c->stCap = want + want / 8 + 1024;
    }
    // ---- resolve the fast kernels' records: per-read merge
    bool mergePending = false;
    if (fast && staged > 0) {
        const ProfLayout PL = prof_layout(nTaxa, nK);
        if ((rc = timer_begin(c, c->timers[KASA_STAGE_SCORE], &a, &b))) return rc;
        hipEvent_t ka, kb;
        if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_ROW_MERGE], &ka, &kb))) return rc;
        uint64_t *noKeys = gp ? nullptr : c->profKeys.as<uint64_t>();   // (narrow records: the rows' events leave no profile keys -- group_stage)
        if (nTaxa <= (uint32_t)BM_WORDS * 32u && !(c->debugFlags & 4)) {
            uint32_t mLo = 0;
            if (nTaxa <= 2048u) {
                // (three sizes: the LDS a row needs is what limits the resident wavefronts of this latency-bound kernel)
                row_merge_bitmap_kernel<256, 64><<<std::min<uint32_t>(nReads, 256u * 32u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(),
                    c->rowLen.as<uint32_t>(), c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, nTaxa, 0u, PL);
                row_merge_bitmap_kernel<512, 64><<<std::min<uint32_t>(nReads, 256u * 32u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(),
                    c->rowLen.as<uint32_t>(), c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, nTaxa, 256u, PL);
                mLo = 512;
            } else {                                                 // up to 16384 taxa: the same two sizes over the wide bitmap
                row_merge_bitmap_kernel<256, BM_WORDS><<<std::min<uint32_t>(nReads, 256u * 32u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(),
                    c->rowLen.as<uint32_t>(), c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, nTaxa, 0u, PL);
                mLo = 256;
            }
            row_merge_bitmap_kernel<RMAX, BM_WORDS><<<std::min<uint32_t>(nReads, 256u * 16u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(),
                c->rowLen.as<uint32_t>(), c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, nTaxa, mLo, PL);
            if (A.rowPerQuery && (uint64_t)c->maxCnt * A.rowPerQuery >= 2ull * RMAX) {   // long reads' rows (beyond RMAX records): streamed, slots in LDS
                if (nTaxa <= 2048u)
                    row_merge_bitmap_kernel<2048, 64, true><<<std::min<uint32_t>(nReads, 256u * 16u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(),
                        c->rowLen.as<uint32_t>(), c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, nTaxa, (uint32_t)RMAX, PL);
                else
                    row_merge_bitmap_kernel<4096, 128, true><<<std::min<uint32_t>(nReads, 256u * 16u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(),
                        c->rowLen.as<uint32_t>(), c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, nTaxa, (uint32_t)RMAX, PL);
            }
        } else
            row_merge_kernel<<<std::min<uint32_t>(nReads, 256u * 24u), 64, 0, c->stream>>>(c->rowPos.as<uint32_t>(), c->rowLen.as<uint32_t>(),
                c->rowKey.as<uint32_t>(), nReads, c->st.as<uint2>(), noKeys, c->kHigh, PL);
        HIPCHK(hipGetLastError());
        if ((rc = timer_end(c, c->kernels[KASA_KERNEL_ROW_MERGE], ka, kb))) return rc;
        c->lastStaged = staged;
        if (!gp) {
            c->lastKeys = nKeys;
            if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_PROFILE_TABLES], &ka, &kb))) return rc;
            if ((rc = profile_from_keys(c, nKeys, false))) return rc;
            if ((rc = timer_end(c, c->kernels[KASA_KERNEL_PROFILE_TABLES], ka, kb))) return rc;
        }
        mergePending = true;
    }
    if (wantPerRead) {
        // CSR offsets = exclusive scan of the row lengths, then rows copied in read order
        if (!mergePending && (rc = timer_begin(c, c->timers[KASA_STAGE_SCORE], &a, &b))) return rc;
        DevBuf &len64 = c->qReadA; // reuse
        if ((rc = len64.reserve(((size_t)nReads + 1) * 8 + 64))) return rc;
        hipEvent_t ca, cb;
        if ((rc = timer_begin(c, c->kernels[KASA_KERNEL_ROW_COPY], &ca, &cb))) return rc;
        HIPCHK(hipMemsetAsync(len64.p, 0, ((size_t)nReads + 1) * 8, c->stream));
        widen_kernel<<<blocks_for(nReads, 256), 256, 0, c->stream>>>(c->rowLen.as<uint32_t>(), len64.as<uint64_t>(), nReads);
        size_t tmpBytes = 0;
        HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, len64.as<uint64_t>(), c->rowOff.as<uint64_t>(), (uint64_t)0, (size_t)nReads + 1,
                                       rocprim::plus<uint64_t>(), c->stream));
        if ((rc = c->scanTmp.reserve(tmpBytes))) return rc;              // (not sortTmp: the profile's leftover sort is queued with that)
        HIPCHK(rocprim::exclusive_scan(c->scanTmp.p, tmpBytes, len64.as<uint64_t>(), c->rowOff.as<uint64_t>(), (uint64_t)0, (size_t)nReads + 1,
                                       rocprim::plus<uint64_t>(), c->stream));
        uint64_t nnz = 0;
        HIPCHK(hipMemcpyAsync(&nnz, c->rowOff.as<uint64_t>() + nReads, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->nnz = nnz;
        // The rows stay where the score kernels left them ({taxon, score} pairs in the staging buffer, row r at rowPos[r], rowOff
        // = the running sum of their lengths): kasa_batch_rank reads them there.  Only a host that asks for the whole CSR
        // (kasa_batch_scores_fetch) has them packed into two arrays first (csr_pack: the copy every batch paid until round 4).
        c->csrPacked = false;
        if ((rc = timer_end(c, c->kernels[KASA_KERNEL_ROW_COPY], ca, cb))) return rc;
    }
    if (mergePending || wantPerRead) { if ((rc = timer_end(c, c->timers[KASA_STAGE_SCORE], a, b))) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (wantPerRead) c->haveScores = true;
    c->state = 4;
    return KASA_OK;
}


extern "C" int kasa_batch_lookup_score(kasa_ctx *c, int wantPerRead, int coverage)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_lookup_score: batch not sorted");
    int rc = group_stage(c, coverage, false);
    if (rc) return rc;
    if (c->recWords() == 8 && !wantPerRead && !c->forceSlowScore) {                  // kASA without -q: the profile is complete after the group stage
        c->haveScores = false; c->nnz = 0; c->state = 4;
        return KASA_OK;
    }
    return score_stage(c, wantPerRead);
}

extern "C" int kasa_batch_group(kasa_ctx *c, int coverage)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    return group_stage(c, coverage, true);
}

extern "C" int kasa_batch_group_to(kasa_ctx *c, int coverage, uint32_t *recordsOut)
{
    if (!c || !recordsOut) return fail(KASA_E_ARG, "kasa_batch_group_to: NULL argument");
    return group_stage(c, coverage, true, recordsOut);
}

extern "C" int kasa_batch_score(kasa_ctx *c, int wantPerRead)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    return score_stage(c, wantPerRead);
}

extern "C" int kasa_batch_records_size(kasa_ctx *c, uint64_t *nRecordWords, uint64_t *nPoolWords)
{
    if (!c || !nRecordWords || !nPoolWords) return fail(KASA_E_ARG, "kasa_batch_records_size: NULL argument");
    if (!c->grouped || !c->recSorted) return fail(KASA_E_STATE, "kasa_batch_records_size: no exported event records (call kasa_batch_group)");
    *nRecordWords = c->nQ * (uint64_t)c->recWords();
    *nPoolWords = c->poolUsed;
    return KASA_OK;
}

extern "C" int kasa_batch_records_fetch(kasa_ctx *c, uint32_t *records, uint32_t *pool)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (!c->grouped || !c->recSorted) return fail(KASA_E_STATE, "kasa_batch_records_fetch: no exported event records (call kasa_batch_group)");
    HIPCHK(hipSetDevice(c->ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint64_t n = c->nQ * (uint64_t)c->recWords();
    // (kasa_batch_group_to wrote them into the caller's buffer, not into `rec`)
    if (n && records) HIPCHK(hipMemcpy(records, c->recOut ? (const void *)c->recOut : c->rec.p, n * 4, hipMemcpyDeviceToHost));
    if (pool && c->poolUsed) {
        if (c->nQ) HIPCHK(hipMemcpy(pool, c->pool.p, (size_t)c->poolUsed * 4, hipMemcpyDeviceToHost));
        pool[0] = 0;                                                   // word 0 is the cursor's start: never referenced
    }
    return KASA_OK;
}

// records in sorted order -> their slots; the depth of every sorted position comes with them (the importing context
// did not look these queries up itself)
template <int RW>
__global__ void place_records_kernel(const uint4 *__restrict__ in, const uint32_t *__restrict__ slotOf, uint32_t n, uint4 *__restrict__ rec,
                                     uint8_t *__restrict__ depth)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t s = slotOf[p];
    depth[p] = (uint8_t)(in[(size_t)p * (RW / 4)].z & 31u);
#pragma unroll
    for (int w = 0; w < RW / 4; ++w) rec[(size_t)s * (RW / 4) + w] = in[(size_t)p * (RW / 4) + w];
}

// first closing position of every tile and level from the sorted queries and their depths (what the lookup kernels
// emit on the way; needed again when the depths were imported)
template <class Key>
__global__ __launch_bounds__(TILE_THREADS) void tile_first_kernel(const Key *__restrict__ qKmer, const uint8_t *__restrict__ depth, uint32_t nQ,
                                                                  int kHigh, int kLow, uint32_t *__restrict__ tileFirst, uint32_t nTiles)
{
    __shared__ uint32_t sFirst[MAX_LEVELS];
    const int nK = kHigh - kLow + 1;
    const uint32_t allLv = (nK >= 32) ? 0xFFFFFFFFu : ((1u << nK) - 1u);
    if (threadIdx.x < MAX_LEVELS) sFirst[threadIdx.x] = NOPOS;
    __syncthreads();
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t p = blockIdx.x * TILE + it * TILE_THREADS + threadIdx.x;
        if (p >= nQ) continue;
        const int ql = (p == 0) ? 0 : lcp_letters<Key>(qKmer[p - 1], qKmer[p]);
        const uint32_t m = special_mask(ql, (int)depth[p], kHigh, allLv);
        for (int lv = 0; lv < nK; ++lv) if ((m >> lv) & 1u) atomicMin(&sFirst[lv], p);
    }
    __syncthreads();
    if ((int)threadIdx.x < nK) tileFirst[(size_t)threadIdx.x * nTiles + blockIdx.x] = sFirst[threadIdx.x];
}

static int import_tail(kasa_ctx *c, uint64_t nRecordWords, uint64_t nPoolWords);

extern "C" int kasa_batch_records_import(kasa_ctx *c, const uint32_t *records, uint64_t nRecordWords, const uint32_t *pool, uint64_t nPoolWords)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_records_import: batch not sorted");
    const int RW = c->recWords();
    if (nRecordWords != c->nQ * (uint64_t)RW) return fail(KASA_E_ARG, "kasa_batch_records_import: %llu record words for %llu queries x %d words", (unsigned long long)nRecordWords, (unsigned long long)c->nQ, RW);
    if ((nRecordWords && !records) || (nPoolWords && !pool) || nPoolWords >= 0xFFFFFFF0ull) return fail(KASA_E_ARG, "kasa_batch_records_import: bad arguments");
    HIPCHK(hipSetDevice(c->ix->device));
    int rc;
    if (c->nQ && !c->slotOf && (rc = slots_from_reads(c))) return rc;
    if ((rc = c->rec.reserve(nRecordWords * 4 + 64)) || (rc = c->recIn.reserve(nRecordWords * 4 + 64)) || (rc = c->pool.reserve((nPoolWords + 1) * 4))) return rc;
    c->recCW = (uint32_t)RW;
    c->poolCap = std::max<uint64_t>(c->poolCap, nPoolWords + 1);
    if (nRecordWords) HIPCHK(hipMemcpyAsync(c->recIn.p, records, nRecordWords * 4, hipMemcpyHostToDevice, c->stream));
    if (nPoolWords) HIPCHK(hipMemcpyAsync(c->pool.p, pool, nPoolWords * 4, hipMemcpyHostToDevice, c->stream));
    return import_tail(c, nRecordWords, nPoolWords);
}

// recIn (sorted order) and pool are in place: file the records by slot, rebuild depths and the per-tile tables
static int import_tail(kasa_ctx *c, uint64_t nRecordWords, uint64_t nPoolWords)
{
    const int RW = c->recWords();
    if (nRecordWords) {
        if (RW == 8) place_records_kernel<8><<<blocks_for(c->nQ, 256), 256, 0, c->stream>>>(c->recIn.as<uint4>(), c->slotOf, (uint32_t)c->nQ, c->rec.as<uint4>(), c->depth.as<uint8_t>());
        else place_records_kernel<16><<<blocks_for(c->nQ, 256), 256, 0, c->stream>>>(c->recIn.as<uint4>(), c->slotOf, (uint32_t)c->nQ, c->rec.as<uint4>(), c->depth.as<uint8_t>());
        const uint32_t nTiles = (uint32_t)((c->nQ + TILE - 1) / TILE);
        if (c->ix->wide) tile_first_kernel<key128><<<nTiles, TILE_THREADS, 0, c->stream>>>(c->keys<key128>(), c->depth.as<uint8_t>(), (uint32_t)c->nQ, c->kHigh, c->kLow, c->tileFirst.as<uint32_t>(), nTiles);
        else tile_first_kernel<uint64_t><<<nTiles, TILE_THREADS, 0, c->stream>>>(c->keys<uint64_t>(), c->depth.as<uint8_t>(), (uint32_t)c->nQ, c->kHigh, c->kLow, c->tileFirst.as<uint32_t>(), nTiles);
        { int rc2 = tile_suffix(c, nTiles); if (rc2) return rc2; }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->poolUsed = (uint32_t)std::max<uint64_t>(1, nPoolWords);
    c->grouped = true; c->recSorted = false; c->haveScores = false;
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// device-resident exchange for the range-partitioned index (C5): slices of sorted queries and the event records made
// from them move between contexts (devices) as device pointers; the caller's collective (RCCL all_to_all) carries them
// ------------------------------------------------------------------------------------------------
extern "C" int kasa_batch_queries_device(kasa_ctx *c, const void **kmers, uint64_t *n)
{
    if (!c || !kmers || !n) return fail(KASA_E_ARG, "kasa_batch_queries_device: NULL argument");
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_queries_device: batch not sorted");
    HIPCHK(hipSetDevice(c->ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    *kmers = c->qKmer; *n = c->nQ;
    return KASA_OK;
}

template <class Key>
__global__ void slice_starts_kernel(const Key *__restrict__ qKmer, uint32_t nQ, const uint64_t *__restrict__ cuts, uint32_t nParts, int shift,
                                    uint64_t *__restrict__ starts)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nParts) return;
    if (j == nParts) { starts[j] = nQ; return; }
    if (j == 0) { starts[0] = 0; return; }
    uint32_t lo = 0, hi = nQ;                                          // first query whose 30-bit prefix is >= cuts[j]
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)(qKmer[mid] >> shift) < cuts[j]) lo = mid + 1; else hi = mid; }
    starts[j] = lo;
}

extern "C" int kasa_batch_slice_starts(kasa_ctx *c, const uint64_t *cuts, uint32_t nParts, uint64_t *starts)
{
    if (!c || !cuts || !starts || nParts == 0) return fail(KASA_E_ARG, "kasa_batch_slice_starts: bad arguments");
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_slice_starts: batch not sorted");
    HIPCHK(hipSetDevice(c->ix->device));
    ScopedBuf tmp;
    int rc = tmp.reserve(((size_t)nParts * 2 + 1) * 8);
    if (rc) return rc;
    uint64_t *dCuts = tmp.as<uint64_t>(), *dStarts = dCuts + nParts;
    HIPCHK(hipMemcpyAsync(dCuts, cuts, (size_t)nParts * 8, hipMemcpyHostToDevice, c->stream));
    const int shift = 5 * (c->K() - RANGE_LETTERS);
    if (c->ix->wide) slice_starts_kernel<key128><<<blocks_for(nParts + 1, 64), 64, 0, c->stream>>>(c->keys<key128>(), (uint32_t)c->nQ, dCuts, nParts, shift, dStarts);
    else slice_starts_kernel<uint64_t><<<blocks_for(nParts + 1, 64), 64, 0, c->stream>>>(c->keys<uint64_t>(), (uint32_t)c->nQ, dCuts, nParts, shift, dStarts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(starts, dStarts, ((size_t)nParts + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

extern "C" int kasa_batch_set_sorted_device(kasa_ctx *c, const void *kmersDev, uint64_t n)
{
    if (!c || (n && !kmersDev)) return fail(KASA_E_ARG, "kasa_batch_set_sorted_device: bad arguments");
    if (n >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_batch_set_sorted_device: too many queries for one batch");
    HIPCHK(hipSetDevice(c->ix->device));
    c->state = 0; c->haveScores = false; c->grouped = false; c->slotOf = nullptr; c->payloadIsSlot = false; c->rankValid = false; c->txtValid = false; c->cohScores = nullptr;
    c->recOut = nullptr; c->recSorted = false;                       // the last batch's exported records (the caller's buffer) are not this batch's
    int rc;
    if ((rc = c->qKmerB.reserve(n * c->keyBytes() + 64))) return rc;
    if (n) HIPCHK(hipMemcpyAsync(c->qKmerB.p, kmersDev, n * c->keyBytes(), hipMemcpyDefault, c->stream));   // same device or a peer's memory
    c->nReads = 0; c->nSeq = 0; c->nQ = n; c->maxCnt = 0; c->readsUploaded = false;
    c->qKmer = c->qKmerB.p; c->qRead = nullptr;
    rc = c->ix->wide ? lookup_part<key128>(c) : lookup_part<uint64_t>(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

extern "C" int kasa_batch_records_device(kasa_ctx *c, const uint32_t **records, uint64_t *nRecordWords, const uint32_t **pool, uint64_t *nPoolWords)
{
    if (!c || !records || !nRecordWords || !pool || !nPoolWords) return fail(KASA_E_ARG, "kasa_batch_records_device: NULL argument");
    if (!c->grouped || !c->recSorted) return fail(KASA_E_STATE, "kasa_batch_records_device: no exported event records (call kasa_batch_group)");
    HIPCHK(hipSetDevice(c->ix->device));
    if (!c->pool.p) {                                                     // an empty slice never reserved the pool: word 0 (never referenced) must still exist
        int rc = c->pool.reserve(64);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(c->pool.p, 0, 64, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    *records = c->recOut ? c->recOut : c->rec.as<uint32_t>(); *nRecordWords = c->nQ * (uint64_t)c->recWords();
    *pool = c->pool.as<uint32_t>(); *nPoolWords = c->poolUsed;
    return KASA_OK;
}

// records of one slice into the batch: sorted positions move by the slice start, pool offsets by the pool base
// (in == out when the records were received in the inbox itself: a thread reads its whole record before it writes it)
template <int RW>
__global__ void shift_records_kernel(const uint4 *in, uint32_t n, uint32_t start, uint32_t poolShift, uint4 *out)
{
    typedef RecTraits<RW> RT;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint4 v[RW / 4];
#pragma unroll
    for (int w = 0; w < RW / 4; ++w) v[w] = in[(size_t)p * (RW / 4) + w];
    const bool matched = (v[0].z & 31u) != 0u;
    v[0].x += start;
    if (matched) v[0].y += start;
    const uint32_t nseg = RW == 8 ? (v[0].w & 255u) : v[0].w;
    if (matched && nseg > (uint32_t)RT::INL) v[RW / 4 - 1].w += poolShift;
#pragma unroll
    for (int w = 0; w < RW / 4; ++w) out[(size_t)p * (RW / 4) + w] = v[w];
}

// ---- the records of a slice, packed for the wire (the return leg of the partitioned exchange).  A record of an exported
// slice says little that its receiver does not know: word [0] is the query's place in the slice, an unmatched query's record
// is empty, and most matched queries use one to three of their four (eight) segment words.  On the wire: one byte of
// CLASSES per four records (2 bits each: 0 = unmatched, nothing follows; 1, 2, 3 = the record's words [1 .. n] with
// n = 4 / 6 / 7 for 32-byte records -- up to one / up to three / more segments -- and 9 / 11 / 15 for 64-byte ones: up to two /
// up to four / more), padded to 16 bytes, then the words back to back in slice order.  SURVEY 8(e) sizes the exchange at
// 12-20 bytes per query; whole records are 32 / 64.
template <int RW> __device__ __forceinline__ uint32_t wire_class(const uint32_t *w)
{
    if ((w[2] & 31u) == 0u) return 0u;
    const uint32_t n = RW == 8 ? (w[3] & 255u) : w[3];
    return RW == 8 ? (n <= 1u ? 1u : (n <= 3u ? 2u : 3u)) : (n <= 2u ? 1u : (n <= 4u ? 2u : 3u));
}
template <int RW> __device__ __forceinline__ uint32_t wire_words(uint32_t cls)
{
    return RW == 8 ? (cls == 0u ? 0u : cls == 1u ? 4u : cls == 2u ? 6u : 7u) : (cls == 0u ? 0u : cls == 1u ? 9u : cls == 2u ? 11u : 15u);
}
static constexpr int WIRE_BLOCK = 1024;                               // records per workgroup (256 threads, four records each)
// words the records of every block put on the wire (pack: classes from the records; unpack: from the class bytes)
template <int RW, bool FROM_RECORDS>
__global__ __launch_bounds__(256) void wire_count_kernel(const uint32_t *__restrict__ rec, const uint8_t *__restrict__ classes, uint64_t n, uint64_t *__restrict__ blockWords)
{
    __shared__ uint32_t sh[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    uint32_t words = 0;
    if (FROM_RECORDS) { for (int q = 0; q < 4; ++q) if (i0 + q < n) words += wire_words<RW>(wire_class<RW>(rec + (i0 + q) * RW)); }
    else if (i0 < n) { const uint32_t cb = classes[i0 >> 2]; for (int q = 0; q < 4; ++q) if (i0 + q < n) words += wire_words<RW>((cb >> (2 * q)) & 3u); }
    words = wave_incl_sum(words);
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = words;
    __syncthreads();
    if (threadIdx.x == 0) blockWords[blockIdx.x] = (uint64_t)sh[0] + sh[1] + sh[2] + sh[3];
}
template <int RW>
__global__ __launch_bounds__(256) void wire_pack_kernel(const uint32_t *__restrict__ rec, uint64_t n, const uint64_t *__restrict__ blockOff, uint8_t *__restrict__ classes,
                                                        uint32_t *__restrict__ words)
{
    __shared__ uint32_t sh[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    uint32_t cls[4] = {0u, 0u, 0u, 0u}, mine = 0;
    for (int q = 0; q < 4; ++q) if (i0 + q < n) { cls[q] = wire_class<RW>(rec + (i0 + q) * RW); mine += wire_words<RW>(cls[q]); }
    if (i0 < n) classes[i0 >> 2] = (uint8_t)(cls[0] | (cls[1] << 2) | (cls[2] << 4) | (cls[3] << 6));
    const uint32_t incl = wave_incl_sum(mine);
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += sh[w];
    uint32_t *o = words + blockOff[blockIdx.x] + before;
    for (int q = 0; q < 4; ++q) {
        const uint32_t nw = wire_words<RW>(cls[q]);
        const uint32_t *w = rec + (i0 + q) * RW;
        for (uint32_t x = 0; x < nw; ++x) *o++ = w[1u + x];
    }
}
template <int RW>
__global__ __launch_bounds__(256) void wire_unpack_kernel(const uint8_t *__restrict__ classes, const uint32_t *__restrict__ words, uint64_t n, const uint64_t *__restrict__ blockOff,
                                                          uint32_t *__restrict__ rec)
{
    __shared__ uint32_t sh[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    uint32_t cls[4] = {0u, 0u, 0u, 0u}, mine = 0;
    if (i0 < n) { const uint32_t cb = classes[i0 >> 2]; for (int q = 0; q < 4; ++q) if (i0 + q < n) { cls[q] = (cb >> (2 * q)) & 3u; mine += wire_words<RW>(cls[q]); } }
    const uint32_t incl = wave_incl_sum(mine);
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += sh[w];
    const uint32_t *in = words + blockOff[blockIdx.x] + before;
    for (int q = 0; q < 4; ++q) {
        if (i0 + q >= n) break;
        uint32_t *w = rec + (i0 + q) * RW;
        const uint32_t nw = wire_words<RW>(cls[q]);
        w[0] = (uint32_t)(i0 + q);                                     // the query's place in its slice
        for (uint32_t x = 0; x < (uint32_t)RW - 1u; ++x) w[1u + x] = x < nw ? *in++ : 0u;
    }
}
static inline uint64_t wire_class_bytes(uint64_t n) { return ((n + 3) / 4 + 15) / 16 * 16; }
// block offsets (a running sum over the blocks' word counts, in ctx->wireOff) -> total words
template <bool FROM_RECORDS>
static int wire_offsets(kasa_ctx *c, const uint32_t *rec, const uint8_t *classes, uint64_t n, uint64_t *totalWords)
{
    int rc;
    const uint64_t nBlocks = (n + WIRE_BLOCK - 1) / WIRE_BLOCK;
    *totalWords = 0;
    if (n == 0) return KASA_OK;
    if ((rc = c->wireOff.reserve((nBlocks + 1) * 8 + 64))) return rc;
    uint64_t *off = c->wireOff.as<uint64_t>();
    HIPCHK(hipMemsetAsync(off + nBlocks, 0, 8, c->stream));
    if (c->recWords() == 8) wire_count_kernel<8, FROM_RECORDS><<<(unsigned)nBlocks, 256, 0, c->stream>>>(rec, classes, n, off);
    else wire_count_kernel<16, FROM_RECORDS><<<(unsigned)nBlocks, 256, 0, c->stream>>>(rec, classes, n, off);
    size_t tmpBytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, off, off, (uint64_t)0, (size_t)nBlocks + 1, rocprim::plus<uint64_t>(), c->stream));
    if ((rc = c->scanTmp.reserve(tmpBytes))) return rc;
    HIPCHK(rocprim::exclusive_scan(c->scanTmp.p, tmpBytes, off, off, (uint64_t)0, (size_t)nBlocks + 1, rocprim::plus<uint64_t>(), c->stream));
    HIPCHK(hipMemcpyAsync(totalWords, off + nBlocks, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

extern "C" int kasa_batch_records_pack_size(kasa_ctx *c, const uint32_t *recordsDev, uint64_t nQueries, uint64_t *nBytes)
{
    if (!c || !nBytes || (nQueries && !recordsDev)) return fail(KASA_E_ARG, "kasa_batch_records_pack_size: NULL argument");
    if (nQueries >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_batch_records_pack_size: a slice holds fewer than 2^32 queries");
    HIPCHK(hipSetDevice(c->ix->device));
    uint64_t words = 0;
    int rc = wire_offsets<true>(c, recordsDev, nullptr, nQueries, &words);
    if (rc) return rc;
    c->wireQueries = nQueries; c->wireRecords = recordsDev; c->wireWords = words;
    *nBytes = wire_class_bytes(nQueries) + words * 4;
    return KASA_OK;
}

extern "C" int kasa_batch_records_pack(kasa_ctx *c, const uint32_t *recordsDev, uint64_t nQueries, void *outDev, uint64_t capBytes)
{
    if (!c || (nQueries && (!recordsDev || !outDev))) return fail(KASA_E_ARG, "kasa_batch_records_pack: NULL argument");
    if (c->wireQueries != nQueries || c->wireRecords != recordsDev) return fail(KASA_E_STATE, "kasa_batch_records_pack: call kasa_batch_records_pack_size for these records first");
    if (wire_class_bytes(nQueries) + c->wireWords * 4 > capBytes) return fail(KASA_E_ARG, "kasa_batch_records_pack: the buffer is too small");
    if (nQueries == 0) return KASA_OK;
    HIPCHK(hipSetDevice(c->ix->device));
    const uint64_t nBlocks = (nQueries + WIRE_BLOCK - 1) / WIRE_BLOCK;
    uint8_t *classes = static_cast<uint8_t *>(outDev);
    uint32_t *words = reinterpret_cast<uint32_t *>(classes + wire_class_bytes(nQueries));
    if (wire_class_bytes(nQueries) > (nQueries + 3) / 4)                  // (the padding of the class bytes: zeros on the wire)
        HIPCHK(hipMemsetAsync(classes + (nQueries + 3) / 4, 0, wire_class_bytes(nQueries) - (nQueries + 3) / 4, c->stream));
    if (c->recWords() == 8) wire_pack_kernel<8><<<(unsigned)nBlocks, 256, 0, c->stream>>>(recordsDev, nQueries, c->wireOff.as<uint64_t>(), classes, words);
    else wire_pack_kernel<16><<<(unsigned)nBlocks, 256, 0, c->stream>>>(recordsDev, nQueries, c->wireOff.as<uint64_t>(), classes, words);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->wireRecords = nullptr;
    return KASA_OK;
}

extern "C" int kasa_batch_records_unpack(kasa_ctx *c, const void *packedDev, uint64_t nBytes, uint64_t nQueries, uint32_t *recordsOutDev)
{
    if (!c || (nQueries && (!packedDev || !recordsOutDev))) return fail(KASA_E_ARG, "kasa_batch_records_unpack: NULL argument");
    if (nQueries == 0) return KASA_OK;
    if (nBytes < wire_class_bytes(nQueries)) return fail(KASA_E_ARG, "kasa_batch_records_unpack: %llu bytes cannot hold the classes of %llu queries", (unsigned long long)nBytes, (unsigned long long)nQueries);
    HIPCHK(hipSetDevice(c->ix->device));
    const uint8_t *classes = static_cast<const uint8_t *>(packedDev);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(classes + wire_class_bytes(nQueries));
    uint64_t total = 0;
    int rc = wire_offsets<false>(c, nullptr, classes, nQueries, &total);
    if (rc) return rc;
    if (wire_class_bytes(nQueries) + total * 4 != nBytes) return fail(KASA_E_ARG, "kasa_batch_records_unpack: the classes announce %llu words, the buffer holds %llu bytes", (unsigned long long)total, (unsigned long long)nBytes);
    const uint64_t nBlocks = (nQueries + WIRE_BLOCK - 1) / WIRE_BLOCK;
    if (c->recWords() == 8) wire_unpack_kernel<8><<<(unsigned)nBlocks, 256, 0, c->stream>>>(classes, words, nQueries, c->wireOff.as<uint64_t>(), recordsOutDev);
    else wire_unpack_kernel<16><<<(unsigned)nBlocks, 256, 0, c->stream>>>(classes, words, nQueries, c->wireOff.as<uint64_t>(), recordsOutDev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

extern "C" int kasa_batch_records_inbox(kasa_ctx *c, uint64_t nRecordWords, uint32_t **records)
{
    if (!c || !records) return fail(KASA_E_ARG, "kasa_batch_records_inbox: NULL argument");
    HIPCHK(hipSetDevice(c->ix->device));
    int rc = c->recIn.reserve(nRecordWords * 4 + 64);
    if (rc) return rc;
    *records = c->recIn.as<uint32_t>();
    return KASA_OK;
}

extern "C" int kasa_batch_records_import_device(kasa_ctx *c, uint32_t nParts, const uint32_t *const *records, const uint64_t *nRecordWords,
                                                const uint32_t *const *pool, const uint64_t *nPoolWords)
{
    if (!c || !records || !nRecordWords || !pool || !nPoolWords) return fail(KASA_E_ARG, "kasa_batch_records_import_device: NULL argument");
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_records_import_device: batch not sorted");
    const int RW = c->recWords();
    uint64_t totalRec = 0, totalPool = 1;                                 // pool word 0 is never referenced
    for (uint32_t j = 0; j < nParts; ++j) {
        if (nRecordWords[j] % (uint64_t)RW) return fail(KASA_E_ARG, "kasa_batch_records_import_device: part %u is not a whole number of records", j);
        totalRec += nRecordWords[j];
        totalPool += nPoolWords[j] ? nPoolWords[j] - 1 : 0;
    }
    if (totalRec != c->nQ * (uint64_t)RW) return fail(KASA_E_ARG, "kasa_batch_records_import_device: %llu record words for %llu queries x %d words", (unsigned long long)totalRec, (unsigned long long)c->nQ, RW);
    if (totalPool >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_batch_records_import_device: pool exceeds 2^32 words");
    HIPCHK(hipSetDevice(c->ix->device));
    int rc;
    if (c->nQ && !c->slotOf && (rc = slots_from_reads(c))) return rc;
    if ((rc = c->rec.reserve(totalRec * 4 + 64)) || (rc = c->recIn.reserve(totalRec * 4 + 64)) || (rc = c->pool.reserve((totalPool + 1) * 4))) return rc;
    c->recCW = (uint32_t)RW;
    c->poolCap = std::max<uint64_t>(c->poolCap, totalPool + 1);
    uint64_t start = 0, base = 1;
    HIPCHK(hipMemsetAsync(c->pool.p, 0, 4, c->stream));
    for (uint32_t j = 0; j < nParts; ++j) {
        const uint64_t n = nRecordWords[j] / (uint64_t)RW;
        if (n) {
            if (!records[j]) return fail(KASA_E_ARG, "kasa_batch_records_import_device: part %u has no records pointer", j);
            uint4 *out = c->recIn.as<uint4>() + start * (RW / 4);
            if (RW == 8) shift_records_kernel<8><<<blocks_for(n, 256), 256, 0, c->stream>>>(reinterpret_cast<const uint4 *>(records[j]), (uint32_t)n, (uint32_t)start, (uint32_t)(base - 1), out);
            else shift_records_kernel<16><<<blocks_for(n, 256), 256, 0, c->stream>>>(reinterpret_cast<const uint4 *>(records[j]), (uint32_t)n, (uint32_t)start, (uint32_t)(base - 1), out);
            HIPCHK(hipGetLastError());
        }
        if (nPoolWords[j] > 1) {
            if (!pool[j]) return fail(KASA_E_ARG, "kasa_batch_records_import_device: part %u has no pool pointer", j);
            HIPCHK(hipMemcpyAsync(c->pool.as<uint32_t>() + base, pool[j] + 1, (nPoolWords[j] - 1) * 4, hipMemcpyDefault, c->stream));
            base += nPoolWords[j] - 1;
        }
        start += n;
    }
    return import_tail(c, totalRec, totalPool);
}

extern "C" int kasa_batch_scores_size(kasa_ctx *c, uint64_t *nnz)
{
    if (!c || !nnz) return fail(KASA_E_ARG, "kasa_batch_scores_size: NULL argument");
    if (!c->haveScores) return fail(KASA_E_STATE, "kasa_batch_scores_size: no per-read scores (call kasa_batch_lookup_score with wantPerRead)");
    *nnz = c->nnz;
    return KASA_OK;
}

// the batch's rows packed into the CSR arrays (read order), once per batch and only when somebody wants them
static int csr_pack(kasa_ctx *c)
{
    if (c->csrPacked || c->nnz == 0 || c->nReads == 0) { c->csrPacked = true; return KASA_OK; }
    int rc;
    if ((rc = c->outTax.reserve(c->nnz * 4 + 64)) || (rc = c->outScore.reserve(c->nnz * 4 + 64))) return rc;
    const uint32_t nReads = (uint32_t)c->nReads;
    row_copy_kernel<<<std::min<unsigned>(blocks_for(nReads, 4), 256u * 8u), 256, 0, c->stream>>>(c->rowPos.as<uint32_t>(), c->rowLen.as<uint32_t>(), c->rowOff.as<uint64_t>(),
        nReads, c->st.as<uint2>(), c->outTax.as<uint32_t>(), c->outScore.as<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->csrPacked = true;
    return KASA_OK;
}

extern "C" int kasa_batch_scores_fetch(kasa_ctx *c, uint64_t *readOffsets, uint32_t *taxIdx, float *score)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (!c->haveScores) return fail(KASA_E_STATE, "kasa_batch_scores_fetch: no per-read scores");
    HIPCHK(hipSetDevice(c->ix->device));
    if (taxIdx || score) { const int rc = csr_pack(c); if (rc) return rc; }
    if (readOffsets) HIPCHK(hipMemcpy(readOffsets, c->rowOff.p, ((size_t)c->nReads + 1) * 8, hipMemcpyDeviceToHost));
    if (c->nnz && taxIdx) HIPCHK(hipMemcpy(taxIdx, c->outTax.p, c->nnz * 4, hipMemcpyDeviceToHost));
    if (c->nnz && score) HIPCHK(hipMemcpy(score, c->outScore.p, c->nnz * 4, hipMemcpyDeviceToHost));
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// ranking on the device: what the per-read file can print, instead of the whole CSR, crosses PCIe
// ------------------------------------------------------------------------------------------------
// Compare::scoringFunc (Compare.hpp:1495-1594, 1721-1754) per read: relative score = k-mer score / (1 + log2(freq * span)),
// threshold, sort by relative score (descending), "top hits" while score / max > 0.8f, "further hits" until -b distinct
// k-mer scores were seen.  The denominators come from the host (libm's log2, one row per distinct read length); the
// division is IEEE double on both sides.  One wavefront per read selects the next-best hit (relative score descending,
// taxon ascending = what a stable sort gives) until neither output format would print another one, and emits that
// prefix.  std::sort is only stable up to 16 elements: a read with more hits whose printed prefix touches a tie in the
// relative score is flagged and ranked by the host from its full row (so is one with more than RANK_ROWS hits or a
// prefix beyond RANK_CAP).
static constexpr int RANK_CAP = 64, RANK_ROWS = 256, RANK_SLAB = 256;
static constexpr int RANK_FIRST = 12;   // leading positions rank_exact_kernel's first attempt makes final (a writer prints 3-6 hits as a rule; more: the whole sort)
// rank_exact_kernel keeps a read's hits in LDS: reads with few hits take little of it, so they go to a launch of their own
// that brings more wavefronts to a CU (classes by hit count: <= 32, <= 64, <= 128, more)
__host__ __device__ inline int rank_exact_class(uint32_t cnt) { return cnt <= 32u ? 0 : cnt <= 64u ? 1 : cnt <= 128u ? 2 : 3; }
struct RankEntry { uint32_t tax; float score; double rel; };
// (rows: where the score stage left them -- {taxon, score} pairs in the staging buffer at rowPos[r]; rowOff = their running sum)
__global__ __launch_bounds__(256) void rank_kernel(const uint64_t *__restrict__ rowOff, const uint32_t *__restrict__ rowPos, const uint2 *__restrict__ rows,
                                                   uint32_t nReads, const double *__restrict__ den, uint32_t nTaxa, const uint32_t *__restrict__ readClass,
                                                   double thr, uint32_t beasts, uint4 *__restrict__ meta, RankEntry *__restrict__ entries,
                                                   unsigned long long cap, unsigned long long *__restrict__ cursor, uint32_t *__restrict__ nFlagged,
                                                   RankEntry *__restrict__ handOver, uint16_t *__restrict__ handKey, uint32_t *__restrict__ nClass)
{
    // the read's hits, compacted (taxon ascending, as in the row): everything after the first pass runs out of LDS
    __shared__ uint32_t sTax[4][RANK_ROWS];
    __shared__ float sScore[4][RANK_ROWS];
    __shared__ double sRel[4][RANK_ROWS];
    __shared__ RankEntry sOut[4][RANK_CAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    // output space comes in slabs of RANK_SLAB entries per wavefront: one atomic per slab, not per read (ten million
    // atomics on one address would take longer than everything else here); what a slab has left when the next read does
    // not fit stays unused
    unsigned long long slabAt = 0;
    uint32_t slabLeft = 0, flaggedMine = 0;
    uint32_t classMine[4] = {0u, 0u, 0u, 0u};                              // flagged reads with <= 32, <= 64, <= 128, more hits
    for (uint32_t r = blockIdx.x * 4u + wv; r < nReads; r += gridDim.x * 4u) {
        const uint64_t lo = rowOff[r];
        const uint32_t m = (uint32_t)(rowOff[r + 1] - lo);
        const uint2 *__restrict__ row = rows + rowPos[r];
        const double *dr = den + (size_t)readClass[r] * nTaxa;
        uint32_t cnt = 0;
        float maxV = 0.0f;
        for (uint32_t c0 = 0; c0 < m; c0 += 64) {
            const uint32_t i = c0 + lane;
            bool ok = false; float sc = 0.0f; uint32_t t = 0; double rel = 0.0;
            if (i < m) {
                const uint2 e = row[i];
                sc = __uint_as_float(e.y); t = e.x;
                rel = (double)sc / dr[t];                                     // Compare.hpp:1506-1511, the denominator is the host's
                ok = sc > 0.0f && rel >= thr;
            }
            const unsigned long long mk = __ballot(ok);
            const uint32_t at = cnt + (uint32_t)__popcll(mk & below);
            if (ok && at < (uint32_t)RANK_ROWS) { sTax[wv][at] = t; sScore[wv][at] = sc; sRel[wv][at] = rel; }
            if (ok) maxV = fmaxf(maxV, sc);
            cnt += (uint32_t)__popcll(mk);
        }
        for (int off = 32; off; off >>= 1) maxV = fmaxf(maxV, __shfl_xor(maxV, off));
        LDS_WAVE_SYNC();
        // selection, in step with the two printing loops (JSON / JSONL / Kraken: top + further hits; TSV: one list)
        uint32_t nOut = 0, top = 0, jJ = 0, jT = 0;
        float beforeJ = 0.0f, beforeT = 0.0f;
        bool topDone = false, doneJ = false, doneT = false, flag = cnt > (uint32_t)RANK_ROWS;
        double lastRel = 0.0; uint32_t lastTax = 0;
        for (uint32_t k = 0; k < cnt && !flag; ++k) {
            // the next hit after (lastRel, lastTax) in the order (relative score descending, taxon ascending)
            bool have = false; double bRel = 0.0; uint32_t bTax = 0xFFFFFFFFu; float bScore = 0.0f;
            for (uint32_t i = lane; i < cnt; i += 64) {
                const double rel = sRel[wv][i];
                const uint32_t t = sTax[wv][i];
                if (k > 0 && !(rel < lastRel || (rel == lastRel && t > lastTax))) continue;   // selected before
                if (!have || rel > bRel || (rel == bRel && t < bTax)) { have = true; bRel = rel; bTax = t; bScore = sScore[wv][i]; }
            }
            // The best of the lanes' candidates: three wavefront-wide reductions in the ALU (DPP) -- over the two halves of the
            // relative score's bit pattern, made to order like the doubles, then over the taxon -- and three register reads from
            // the winning lane.  (The butterfly of 64-bit shuffles through the LDS crossbar that stood here, six rounds of five
            // dependent exchanges per selected hit, was most of this kernel's time.)
            {
                unsigned long long u = (unsigned long long)__double_as_longlong(bRel == 0.0 ? 0.0 : bRel);   // (-0.0 and 0.0 are one value)
                u ^= (u >> 63) ? ~0ull : 0x8000000000000000ull;                // unsigned order = the doubles' order
                const int hi = have ? (int)((uint32_t)(u >> 32) ^ 0x80000000u) : (int)0x80000000;
                const int hiMax = wave_max_int(hi);
                const bool c1 = have && hi == hiMax;
                const int lo = c1 ? (int)((uint32_t)u ^ 0x80000000u) : (int)0x80000000;
                const int loMax = wave_max_int(lo);
                const bool c2 = c1 && lo == loMax;
                const int txMin = __builtin_amdgcn_readlane(wave_incl_min(c2 ? (int)bTax : 0x7fffffff), 63);
                const unsigned long long win = __ballot(c2 && (int)bTax == txMin);
                if (win == 0ull) { flag = true; break; }                        // (cannot happen while k < cnt: left to the exact kernel)
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)win) - 1);   // one lane: a read's hits have distinct taxa
                const unsigned long long rb = (unsigned long long)__double_as_longlong(bRel);
                const uint32_t rLo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rb, src), rHi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rb >> 32), src);
                bRel = __longlong_as_double((long long)(((unsigned long long)rHi << 32) | rLo));
                bScore = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bScore), src));
                bTax = (uint32_t)txMin; have = true;
            }
            // ties: other remaining hits with the same relative score (their order is only defined for stable sorts)
            uint32_t same = 0;
            for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool eq = i < cnt && sRel[wv][i] == bRel && sTax[wv][i] > bTax;
                same += (uint32_t)__popcll(__ballot(eq));
            }
            // would a writer print hit k?
            bool printed = false;
            if (!doneT) { if (jT >= beasts) doneT = true; else { printed = true; if (beforeT != bScore) { beforeT = bScore; ++jT; } } }
            if (!topDone) {
                if (k == 0) { top = 1; printed = true; }
                else if (k < beasts && bScore / maxV > 0.8f) { ++top; printed = true; }
                else { topDone = true; jJ = top; }
            }
            if (topDone && !doneJ) { if (jJ >= beasts) doneJ = true; else { printed = true; if (beforeJ != bScore) { beforeJ = bScore; ++jJ; } } }
            if (!printed) break;
            if (same && cnt > 16u) flag = true;
            if (nOut >= (uint32_t)RANK_CAP) { flag = true; break; }
            if (lane == 0) sOut[wv][nOut] = RankEntry{bTax, bScore, bRel};
            ++nOut;
            lastRel = bRel; lastTax = bTax;
        }
        if (flag) {                                                        // rank_exact_kernel (or the host) ranks this read
            nOut = 0;
            if (handOver && cnt <= (uint32_t)RANK_ROWS)                        // its hits, compacted, where the row lies (16 bytes per cell)
                for (uint32_t i = lane; i < cnt; i += 64) {
                    const double mine = sRel[wv][i];
                    uint32_t larger = 0;                                       // order key: hits with a larger relative score (ties share it)
                    for (uint32_t j = 0; j < cnt; ++j) larger += (sRel[wv][j] > mine) ? 1u : 0u;
                    handOver[lo + i] = RankEntry{sTax[wv][i], sScore[wv][i], mine};
                    handKey[lo + i] = (uint16_t)larger;
                }
        }
        if (nOut > slabLeft) {                                             // uniform
            unsigned long long got = 0;
            if (lane == 0) got = atomicAdd(cursor, (unsigned long long)RANK_SLAB);
            slabAt = __shfl(got, 0);
            slabLeft = (uint32_t)RANK_SLAB;
        }
        const unsigned long long at = slabAt;
        slabAt += nOut; slabLeft -= nOut;
        LDS_WAVE_SYNC();
        if (at + nOut <= cap) for (uint32_t x = lane; x < nOut; x += 64) entries[at + x] = sOut[wv][x];
        if (lane == 0) meta[r] = make_uint4((uint32_t)at, nOut | (flag ? 0x80000000u : 0u), __float_as_uint(maxV), cnt);
        flaggedMine += flag ? 1u : 0u;
        if (flag) {
            const int cl = rank_exact_class(cnt);
#pragma unroll
            for (int x = 0; x < 4; ++x) classMine[x] += cl == x ? 1u : 0u;
        }
        LDS_WAVE_SYNC();
    }
    if (lane == 0 && flaggedMine) {
        atomicAdd(nFlagged, flaggedMine);
#pragma unroll
        for (int x = 0; x < 4; ++x) if (classMine[x]) atomicAdd(&nClass[x], classMine[x]);
    }
}

// The reads rank_kernel left: a tie among more than 16 hits inside the printed prefix (std::sort's own order decides who is
// printed), more hits than its LDS holds, or a prefix beyond RANK_CAP.  One THREAD per such read: the hits compacted into
// scratch that parallels the CSR (relative score, position in the row, id), libstdc++'s std::sort walked over the ids
// (stdsort_order.h: the same comparisons and moves, so tied taxa end up where the reference leaves them), the printing
// loops walked over the result, the prefix written out.  A read stays flagged only where std::sort would switch to its
// heap sort or with 65 536 or more cells.
__global__ void rank_list_kernel(const uint4 *__restrict__ meta, uint32_t nReads, uint32_t *__restrict__ list, uint32_t *__restrict__ listKey,
                                 uint32_t *__restrict__ nList)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 mt = make_uint4(0, 0, 0, 0);
    if (r < nReads) mt = meta[r];
    const bool f = r < nReads && (mt.y >> 31) != 0u;
    const unsigned long long mk = __ballot(f);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && mk) base = atomicAdd(nList, (uint32_t)__popcll(mk));
    base = lane_value<0>(base);
    if (f) {
        const uint32_t at = base + (uint32_t)__popcll(mk & ((1ull << (threadIdx.x & 63)) - 1ull));
        list[at] = r;
        listKey[at] = mt.w < 0xFFFFu ? mt.w : 0xFFFFu;                     // its number of hits: the list is sorted by it (lanes of a wavefront do alike)
    }
}

struct PrintWalk {                                                         // the writers' loops (Compare.hpp:1582-1594, 1721-1754), hit by hit
    uint32_t top = 0, jJ = 0, jT = 0;
    float beforeJ = 0.0f, beforeT = 0.0f;
    bool topDone = false, doneJ = false, doneT = false;
    __device__ __forceinline__ bool step(uint32_t k, float score, float maxV, uint32_t beasts)
    {
        bool printed = false;
        if (!doneT) { if (jT >= beasts) doneT = true; else { printed = true; if (beforeT != score) { beforeT = score; ++jT; } } }
        if (!topDone) {
            if (k == 0) { top = 1; printed = true; }
            else if (k < beasts && score / maxV > 0.8f) { ++top; printed = true; }
            else { topDone = true; jJ = top; }
        }
        if (topDone && !doneJ) { if (jJ >= beasts) doneJ = true; else { printed = true; if (beforeJ != score) { beforeJ = score; ++jJ; } } }
        return printed;
    }
};

// One thread's column of an LDS array ([element][lane]) or a piece of global memory, indexed alike.
template <class T>
struct LaneIds {
    T *p; int stride;
    __device__ __forceinline__ T &operator[](int i) const { return p[(size_t)i * stride]; }
};
static constexpr int RANK_EXACT_LANES = 64;                                 // reads per workgroup
// std::sort's waiting ranges of one thread, in its column of an LDS array: up to 256 elements, so first and last take 9
// bits each and the depth limit (2 log2 n <= 16) five
template <int CAP_>
struct LaneStack {
    static constexpr int CAP = CAP_;
    uint32_t *p;
    __device__ __forceinline__ void put(int at, int first, int last, int depth) { p[(size_t)at * RANK_EXACT_LANES] = (uint32_t)first | ((uint32_t)last << 9) | ((uint32_t)depth << 18); }
    __device__ __forceinline__ void get(int at, int &first, int &last, int &depth) const
    {
        const uint32_t w = p[(size_t)at * RANK_EXACT_LANES];
        first = (int)(w & 511u); last = (int)((w >> 9) & 511u); depth = (int)(w >> 18);
    }
};
__host__ __device__ constexpr int rank_exact_stack_cap(int rows) { int lg = 0; while ((rows >> (lg + 1)) != 0) ++lg; return 2 * lg + 2; }
// BIG: the launch for the reads of the last class with MORE hits than its LDS columns take (they sort in global memory with a
// stack in scratch: 412 bytes per lane that the common form does not carry any more); each form skips the other's reads.
template <int ROWS, bool BIG = false>
__global__ __launch_bounds__(RANK_EXACT_LANES) void rank_exact_kernel(const uint32_t *__restrict__ list, uint32_t nList, const uint64_t *__restrict__ rowOff,
                                                        const uint32_t *__restrict__ rowPos, const uint2 *__restrict__ rows,
                                                        const double *__restrict__ den, uint32_t nTaxa, const uint32_t *__restrict__ readClass,
                                                        double thr, uint32_t beasts, RankEntry *__restrict__ hits, uint16_t *__restrict__ keyS,
                                                        uint16_t *__restrict__ idS, uint4 *__restrict__ meta, RankEntry *__restrict__ entries,
                                                        unsigned long long cap, unsigned long long *__restrict__ cursor, uint32_t *__restrict__ nFlagged)
{
    // The sort is a chain of dependent, scattered accesses to the read's hits: what it compares -- an order key per hit
    // (the number of hits with a larger relative score, from rank_kernel: equal scores, equal keys) -- and the ids live in
    // this thread's columns of two LDS arrays (from global scratch the same took 0.5 s for 3.3 M reads).  Reads with more
    // than RANK_ROWS hits (rank_kernel handed nothing over) compact their row themselves and sort in global memory.
    // ROWS: the most hits a read of this launch has (the list is sorted by that number; the host cuts it into classes)
    static_assert(ROWS <= 256, "LaneStack packs positions into 9 bits");
    constexpr int SCAP = rank_exact_stack_cap(ROWS);
    __shared__ uint8_t shKey[ROWS * RANK_EXACT_LANES], shId[ROWS * RANK_EXACT_LANES];   // (up to 256 hits: positions and order keys fit a byte)
    __shared__ uint32_t shStack[SCAP * RANK_EXACT_LANES];
    const uint32_t x = blockIdx.x * RANK_EXACT_LANES + threadIdx.x;
    const int lane = threadIdx.x;
    const bool have = x < nList;
    uint32_t r = 0, cnt = 0, nOut = 0;
    uint64_t lo = 0;
    float maxV = 0.0f;
    bool ok = false, inLds = true, skip = false;
    if (have) {
        r = list[x];
        lo = rowOff[r];
        const uint4 mt = meta[r];
        cnt = mt.w; maxV = __uint_as_float(mt.z);
        inLds = cnt <= (uint32_t)ROWS;
        skip = inLds == BIG;                                                   // (the other form's read)
    }
    if (have && !skip) {
        const uint32_t m = (uint32_t)(rowOff[r + 1] - lo);
        if (inLds) {
            for (uint32_t i = 0; i < cnt; ++i) { shKey[(size_t)i * RANK_EXACT_LANES + lane] = (uint8_t)keyS[lo + i]; shId[(size_t)i * RANK_EXACT_LANES + lane] = (uint8_t)i; }
        } else if (m < 65536u) {                                           // a long row: compacted here, in place
            const double *dr = den + (size_t)readClass[r] * nTaxa;
            uint32_t n = 0;
            for (uint32_t i = 0; i < m; ++i) {
                const uint2 e = rows[(size_t)rowPos[r] + i];
                const float sc = __uint_as_float(e.y);
                const uint32_t t = e.x;
                const double rel = (double)sc / dr[t];
                if (sc > 0.0f && rel >= thr) { hits[lo + n] = RankEntry{t, sc, rel}; idS[lo + n] = (uint16_t)n; ++n; }
            }
        }
        if (inLds || m < 65536u) {
            const RankEntry *hs = hits + lo;
            // only the hits a writer prints have to be in std::sort's order: the first RANK_FIRST positions, as a rule (else the whole sort).
            // (Two instantiations, so that the LDS columns are reached with LDS instructions, not through flat pointers.)
            auto rankRead = [&](auto ids, auto less, auto stack) {
                int covered = 0;
                ok = stdsort_order(ids, (int)cnt, less, RANK_FIRST, &covered, stack);
                for (int pass = 0; ok && pass < 2; ++pass) {
                    PrintWalk w;                                           // how many hits a writer prints
                    bool more = false;
                    nOut = 0;
                    for (uint32_t k = 0; k < cnt; ++k) {
                        if (k >= (uint32_t)covered) { more = true; break; }
                        if (!w.step(k, hs[ids[(int)k]].score, maxV, beasts)) break;
                        ++nOut;
                    }
                    if (!more) break;
                    for (uint32_t i = 0; i < cnt; ++i) ids[(int)i] = (uint16_t)i;   // it prints beyond what is final: all of it, from the start
                    ok = stdsort_order(ids, (int)cnt, less, (int)cnt, &covered, stack);
                }
            };
            if (inLds) {
                const uint8_t *ky = shKey + lane;
                rankRead(LaneIds<uint8_t>{shId + lane, RANK_EXACT_LANES}, [ky](uint16_t a, uint16_t b) { return ky[(size_t)a * RANK_EXACT_LANES] < ky[(size_t)b * RANK_EXACT_LANES]; },
                         LaneStack<SCAP>{shStack + lane});
            } else if constexpr (BIG)                                      // (only the last class holds reads with more hits than its LDS takes)
                rankRead(LaneIds<uint16_t>{idS + lo, 1}, [hs](uint16_t a, uint16_t b) { return hs[a].rel > hs[b].rel; }, StdsortLocalStack());
        }
    }
    uint32_t incl = nOut;                                                  // one allocation per wavefront
    incl = wave_incl_sum(incl);
    const uint32_t total = __shfl(incl, RANK_EXACT_LANES - 1);
    unsigned long long at = 0;
    if (lane == 0 && total) at = atomicAdd(cursor, (unsigned long long)total);
    at = __shfl(at, 0) + incl - nOut;
    if (have && ok) {
        if (at + nOut <= cap)
            for (uint32_t k = 0; k < nOut; ++k) entries[at + k] = hits[lo + (inLds ? shId[(size_t)k * RANK_EXACT_LANES + lane] : idS[lo + k])];
        meta[r] = make_uint4((uint32_t)at, nOut, __float_as_uint(maxV), cnt);
    }
    const unsigned long long left = __ballot(have && !ok && !skip);
    if (lane == 0 && left) atomicAdd(nFlagged, (uint32_t)__popcll(left));
}

extern "C" int kasa_batch_rank(kasa_ctx *c, const double *den, uint32_t nClasses, const uint32_t *readClass, float threshold, uint32_t beasts,
                               uint64_t *nEntries, uint32_t *nFlagged)
{
    if (!c || !den || !readClass || !nEntries || !nFlagged) return fail(KASA_E_ARG, "kasa_batch_rank: NULL argument");
    if (!c->haveScores) return fail(KASA_E_STATE, "kasa_batch_rank: no per-read scores (call kasa_batch_lookup_score with wantPerRead)");
    if (nClasses == 0) return fail(KASA_E_ARG, "kasa_batch_rank: no denominator rows");
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t nTaxa = c->ix->nTaxa, nReads = (uint32_t)c->nReads;
    {   // the class ids index den[] here and bestScore[] in kasa_batch_text: an id out of range would read device memory out of bounds
        uint32_t top = 0;
        for (uint32_t r = 0; r < nReads; ++r) top = readClass[r] > top ? readClass[r] : top;
        if (nReads && top >= nClasses) return fail(KASA_E_ARG, "kasa_batch_rank: readClass names class %u of %u", top, nClasses);
    }
    c->rankClasses = nClasses;
    int rc;
    if ((rc = c->rankDen.reserve((size_t)nClasses * nTaxa * 8)) || (rc = c->rankClass.reserve((size_t)nReads * 4 + 64)) ||
        (rc = c->rankMeta.reserve((size_t)nReads * 16 + 64)))
        return rc;
    HIPCHK(hipMemcpyAsync(c->rankDen.p, den, (size_t)nClasses * nTaxa * 8, hipMemcpyHostToDevice, c->stream));
    if (nReads) HIPCHK(hipMemcpyAsync(c->rankClass.p, readClass, (size_t)nReads * 4, hipMemcpyHostToDevice, c->stream));
    unsigned long long *cursor = c->misc.as<unsigned long long>() + 20;
    uint32_t *flagged = c->misc.as<uint32_t>() + 42, *nClass = c->misc.as<uint32_t>() + 64;   // ([64..67]: flagged reads per class of hit count)
    if (c->rankCap == 0) c->rankCap = std::max<uint64_t>(1 << 20, (uint64_t)nReads * 4 + (uint64_t)RANK_SLAB * 256u * 32u * 4u);
    c->rankEntries = 0; *nEntries = 0; *nFlagged = 0; c->rankValid = false; c->txtValid = false;
    if (nReads == 0) { c->rankValid = true; c->rankFlagged = 0; return KASA_OK; }
    for (int attempt = 0; attempt < 6; ++attempt) {
        if ((rc = c->rankOut.reserve(c->rankCap * sizeof(RankEntry)))) return rc;
        HIPCHK(hipMemsetAsync(cursor, 0, 8, c->stream));
        HIPCHK(hipMemsetAsync(flagged, 0, 4, c->stream));
        HIPCHK(hipMemsetAsync(nClass, 0, 16, c->stream));
        // reads with tied hits are handed to rank_exact_kernel: their compacted hits wait where the row lies (16 bytes per cell)
        const bool exact = c->nnz > 0 && !(c->debugFlags & 256);           // (test tap 256: leave them to the host)
        // (20 bytes per CSR cell.  The event records are dead once the batch is scored: their buffer serves when it is large
        // enough -- 32 bytes per query against 20 per cell -- and the batch then has to be grouped again before another score.)
        DevBuf *scratch = &c->rankScratch;
        if (exact && c->rec.cap >= c->nnz * 20 + 64) { scratch = &c->rec; c->grouped = false; }
        else if (exact && (rc = c->rankScratch.reserve(c->nnz * 20 + 64))) return rc;
        RankEntry *handOver = exact ? scratch->as<RankEntry>() : nullptr;
        uint16_t *handKey = exact ? reinterpret_cast<uint16_t *>(handOver + c->nnz) : nullptr, *idS = handKey ? handKey + c->nnz : nullptr;
        const unsigned blocks = std::min<unsigned>(blocks_for(nReads, 4), 256u * 32u);
        rank_kernel<<<blocks, 256, 0, c->stream>>>(c->rowOff.as<uint64_t>(), c->rowPos.as<uint32_t>(), c->st.as<uint2>(), nReads,
                                                   c->rankDen.as<double>(), nTaxa, c->rankClass.as<uint32_t>(), (double)threshold, beasts,
                                                   c->rankMeta.as<uint4>(), c->rankOut.as<RankEntry>(), c->rankCap, cursor, flagged, handOver, handKey, nClass);
        HIPCHK(hipGetLastError());
        unsigned long long used = 0; uint32_t nf = 0, hClass[4] = {0u, 0u, 0u, 0u};
        HIPCHK(hipMemcpyAsync(&used, cursor, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&nf, flagged, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(hClass, nClass, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (nf > 0 && used <= c->rankCap && exact) {
            // the reads the wavefront-per-read kernel left: std::sort's own order, one thread per read (rank_exact_kernel)
            if ((rc = c->rankList.reserve((size_t)nf * 16 + 64))) return rc;
            uint32_t *list0 = c->rankList.as<uint32_t>(), *key0 = list0 + nf, *list1 = key0 + nf, *key1 = list1 + nf;
            uint32_t *nList = c->misc.as<uint32_t>() + 45;
            HIPCHK(hipMemsetAsync(nList, 0, 4, c->stream));
            HIPCHK(hipMemsetAsync(flagged, 0, 4, c->stream));
            rank_list_kernel<<<blocks_for(nReads, 256), 256, 0, c->stream>>>(c->rankMeta.as<uint4>(), nReads, list0, key0, nList);
            {
                size_t tmpBytes = 0;
                HIPCHK(rocprim::radix_sort_pairs(nullptr, tmpBytes, key0, key1, list0, list1, (size_t)nf, 0u, 16u, c->stream));
                if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
                HIPCHK(rocprim::radix_sort_pairs(c->sortTmp.p, tmpBytes, key0, key1, list0, list1, (size_t)nf, 0u, 16u, c->stream));
            }
            // the list is sorted by the number of hits: one launch per class, the LDS a read's hits take cut to the class
            if (hClass[0] + hClass[1] + hClass[2] + hClass[3] != nf) return fail(KASA_E_HIP, "kasa_batch_rank: the classes of the flagged reads do not add up");
            uint32_t from = 0;
            for (int cl = 0; cl < 4; ++cl) {
                const uint32_t nCl = hClass[cl];
                if (nCl == 0) continue;
#define KASA_RANK_EXACT(...) rank_exact_kernel<__VA_ARGS__><<<blocks_for(nCl, RANK_EXACT_LANES), RANK_EXACT_LANES, 0, c->stream>>>(list1 + from, nCl, c->rowOff.as<uint64_t>(), \
                c->rowPos.as<uint32_t>(), c->st.as<uint2>(), c->rankDen.as<double>(), nTaxa, c->rankClass.as<uint32_t>(), (double)threshold, beasts, \
                handOver, handKey, idS, c->rankMeta.as<uint4>(), c->rankOut.as<RankEntry>(), c->rankCap, cursor, flagged)
                if (c->debugFlags & 1048576) KASA_RANK_EXACT(RANK_ROWS);   // (test tap: every class in the largest form)
                else if (cl == 0) KASA_RANK_EXACT(32);
                else if (cl == 1) KASA_RANK_EXACT(64);
                else if (cl == 2) KASA_RANK_EXACT(128);
                else KASA_RANK_EXACT(RANK_ROWS);
                if (cl == 3 || (c->debugFlags & 1048576)) KASA_RANK_EXACT(RANK_ROWS, true);   // the reads with more hits than the LDS columns take
#undef KASA_RANK_EXACT
                HIPCHK(hipGetLastError());
                from += nCl;
            }
            HIPCHK(hipMemcpyAsync(&used, cursor, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(&nf, flagged, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        if (used <= c->rankCap) { c->rankEntries = used; *nEntries = used; *nFlagged = nf; c->rankValid = true; c->rankFlagged = nf; c->txtValid = false; return KASA_OK; }
        // the kernel has no side effects: grow and rerun.  Output space is handed out in slabs per wavefront, so `used` depends a
        // little on the scheduling: leave a slab for every wavefront that can be in flight on top of the usual slack
        c->rankCap = used + used / 4 + (uint64_t)RANK_SLAB * 256u * 32u * 4u + 1024;
    }
    return fail(KASA_E_LIMIT, "kasa_batch_rank: output did not converge");
}

extern "C" int kasa_batch_rank_fetch(kasa_ctx *c, uint32_t *meta, void *entries)
{
    if (!c || !meta) return fail(KASA_E_ARG, "kasa_batch_rank_fetch: NULL argument");
    if (!c->haveScores) return fail(KASA_E_STATE, "kasa_batch_rank_fetch: no ranked batch");
    HIPCHK(hipSetDevice(c->ix->device));
    if (c->nReads) HIPCHK(hipMemcpyAsync(meta, c->rankMeta.p, (size_t)c->nReads * 16, hipMemcpyDeviceToHost, c->stream));
    if (c->rankEntries && entries) HIPCHK(hipMemcpyAsync(entries, c->rankOut.p, c->rankEntries * sizeof(RankEntry), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// The per-read file's text, written on the device (kasa_text.h)
// ------------------------------------------------------------------------------------------------
extern "C" int kasa_ctx_set_taxa_text(kasa_ctx *c, const uint32_t *taxIds, const char *names, const uint64_t *nameOff)
{
    if (!c || !taxIds || !nameOff) return fail(KASA_E_ARG, "kasa_ctx_set_taxa_text: NULL argument");
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t nTaxa = c->ix->nTaxa;
    for (uint32_t t = 0; t < nTaxa; ++t) if (nameOff[t + 1] < nameOff[t]) return fail(KASA_E_ARG, "kasa_ctx_set_taxa_text: name offsets descend");
    if (nameOff[0] != 0) return fail(KASA_E_ARG, "kasa_ctx_set_taxa_text: name offsets do not start at 0");
    if (nameOff[nTaxa] && !names) return fail(KASA_E_ARG, "kasa_ctx_set_taxa_text: NULL names");
    int rc;
    if ((rc = c->taxText.reserve(nameOff[nTaxa] + 64)) || (rc = c->taxTextOff.reserve(((size_t)nTaxa + 1) * 8)) || (rc = c->taxTextIds.reserve((size_t)nTaxa * 4 + 64))) return rc;
    if (nameOff[nTaxa]) HIPCHK(hipMemcpyAsync(c->taxText.p, names, nameOff[nTaxa], hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->taxTextOff.p, nameOff, ((size_t)nTaxa + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->taxTextIds.p, taxIds, (size_t)nTaxa * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->haveTaxText = true;
    return KASA_OK;
}

extern "C" int kasa_batch_text(kasa_ctx *c, const kasa_text_params *tp, uint64_t *nBytes)
{
    if (!c || !tp || !nBytes) return fail(KASA_E_ARG, "kasa_batch_text: NULL argument");
    if (!c->haveScores || !c->rankValid) return fail(KASA_E_STATE, "kasa_batch_text: the batch is not ranked (kasa_batch_rank)");
    if (c->rankFlagged) return fail(KASA_E_STATE, "kasa_batch_text: kasa_batch_rank left reads to the host; their text is the host's, too");
    if (!c->haveTaxText) return fail(KASA_E_STATE, "kasa_batch_text: no taxon names (kasa_ctx_set_taxa_text)");
    if (tp->format < 0 || tp->format > 3) return fail(KASA_E_ARG, "kasa_batch_text: unknown format");
    if (!tp->readNameOff || !tp->readLen || !tp->bestScore || tp->nClasses == 0) return fail(KASA_E_ARG, "kasa_batch_text: NULL argument");
    if (tp->nClasses != c->rankClasses) return fail(KASA_E_ARG, "kasa_batch_text: %u classes, but kasa_batch_rank was given %u (the class ids of the reads are its)", tp->nClasses, c->rankClasses);
    if (tp->coherence && !c->cohScores) return fail(KASA_E_STATE, "kasa_batch_text: no coherence scores of this batch (kasa_batch_coherence)");
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t nReads = (uint32_t)c->nReads;
    *nBytes = 0; c->txtTotal = 0; c->txtValid = false;
    if (nReads == 0) { c->txtValid = true; return KASA_OK; }
    if (tp->readNameOff[0] != 0) return fail(KASA_E_ARG, "kasa_batch_text: readNameOff does not start at 0");
    for (uint32_t r = 0; r < nReads; ++r)
        if (tp->readNameOff[r + 1] < tp->readNameOff[r]) return fail(KASA_E_ARG, "kasa_batch_text: readNameOff descends at read %u", r);
    const uint64_t nameBytes = tp->readNameOff[nReads];
    if (nameBytes && !tp->readNames) return fail(KASA_E_ARG, "kasa_batch_text: NULL readNames");
    int rc;
    if ((rc = c->txtNames.reserve(nameBytes + 64)) || (rc = c->txtNameOff.reserve(((size_t)nReads + 1) * 8)) || (rc = c->txtLen.reserve((size_t)nReads * 4 + 64)) ||
        (rc = c->txtBest.reserve((size_t)tp->nClasses * 4 + 64)) || (rc = c->txtBytes.reserve(((size_t)nReads + 1) * 8)) || (rc = c->txtOff.reserve(((size_t)nReads + 1) * 8)) ||
        (rc = c->txtFlags.reserve((size_t)nReads + 64)))
        return rc;
    if (nameBytes) HIPCHK(hipMemcpyAsync(c->txtNames.p, tp->readNames, nameBytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->txtNameOff.p, tp->readNameOff, ((size_t)nReads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->txtLen.p, tp->readLen, (size_t)nReads * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->txtBest.p, tp->bestScore, (size_t)tp->nClasses * 4, hipMemcpyHostToDevice, c->stream));
    kasa_text::Args A;
    A.fmt = tp->format; A.beasts = tp->beasts; A.firstRead = tp->firstRead; A.nReads = nReads;
    A.readNames = c->txtNames.as<char>(); A.readNameOff = c->txtNameOff.as<uint64_t>(); A.readLen = c->txtLen.as<uint32_t>();
    A.readClass = c->rankClass.as<uint32_t>(); A.best = c->txtBest.as<float>();
    A.meta = c->rankMeta.as<uint4>(); A.entries = reinterpret_cast<const kasa_text::Entry *>(c->rankOut.p);
    A.taxIds = c->taxTextIds.as<uint32_t>(); A.taxNames = c->taxText.as<char>(); A.taxNameOff = c->taxTextOff.as<uint64_t>();
    A.coherence = tp->coherence ? 1 : 0; A.cohScores = c->cohScores;
    A.errorThreshold = tp->errorThreshold; A.coherenceThreshold = tp->coherenceThreshold;
    static_assert(sizeof(kasa_text::Entry) == sizeof(RankEntry), "kasa_text reads kasa_batch_rank's entries");
    HIPCHK(hipMemsetAsync(c->txtBytes.p, 0, ((size_t)nReads + 1) * 8, c->stream));
    kasa_text::text_size_kernel<<<blocks_for(nReads, 256), 256, 0, c->stream>>>(A, c->txtBytes.as<uint64_t>(), c->txtFlags.as<uint8_t>());
    HIPCHK(hipGetLastError());
    size_t tmpBytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, c->txtBytes.as<uint64_t>(), c->txtOff.as<uint64_t>(), (uint64_t)0, (size_t)nReads + 1, rocprim::plus<uint64_t>(), c->stream));
    if ((rc = c->scanTmp.reserve(tmpBytes))) return rc;
    HIPCHK(rocprim::exclusive_scan(c->scanTmp.p, tmpBytes, c->txtBytes.as<uint64_t>(), c->txtOff.as<uint64_t>(), (uint64_t)0, (size_t)nReads + 1, rocprim::plus<uint64_t>(), c->stream));
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, c->txtOff.as<uint64_t>() + nReads, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = c->txtOut.reserve(total + 64))) return rc;
    kasa_text::text_write_kernel<<<blocks_for(nReads, 64), 64, 0, c->stream>>>(A, c->txtOff.as<uint64_t>(), c->txtOut.as<char>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->txtTotal = total; c->txtValid = true; *nBytes = total;
    return KASA_OK;
}

extern "C" int kasa_batch_text_fetch(kasa_ctx *c, char *text, uint64_t *readOffsets, uint8_t *contaminated)
{
    if (!c) return fail(KASA_E_ARG, "kasa_batch_text_fetch: NULL argument");
    if (!c->txtValid) return fail(KASA_E_STATE, "kasa_batch_text_fetch: no text of this batch (kasa_batch_text)");
    HIPCHK(hipSetDevice(c->ix->device));
    if (c->txtTotal && text) HIPCHK(hipMemcpyAsync(text, c->txtOut.p, c->txtTotal, hipMemcpyDeviceToHost, c->stream));
    if (readOffsets && c->nReads) HIPCHK(hipMemcpyAsync(readOffsets, c->txtOff.p, ((size_t)c->nReads + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    else if (readOffsets) readOffsets[0] = 0;
    if (contaminated && c->nReads) HIPCHK(hipMemcpyAsync(contaminated, c->txtFlags.p, (size_t)c->nReads, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

// a piece of the text: page-locked buffers are expensive to make (seconds for the 5.5 GB of a 10 M-read batch), so a host
// moves the text through a few small ones, writing one to the file while the next arrives
extern "C" int kasa_batch_text_fetch_range(kasa_ctx *c, char *text, uint64_t offset, uint64_t nBytes)
{
    if (!c) return fail(KASA_E_ARG, "kasa_batch_text_fetch_range: NULL argument");
    if (!c->txtValid) return fail(KASA_E_STATE, "kasa_batch_text_fetch_range: no text of this batch (kasa_batch_text)");
    if (offset > c->txtTotal || nBytes > c->txtTotal - offset) return fail(KASA_E_ARG, "kasa_batch_text_fetch_range: beyond the text (%llu bytes)", (unsigned long long)c->txtTotal);
    if (nBytes && !text) return fail(KASA_E_ARG, "kasa_batch_text_fetch_range: NULL text");
    HIPCHK(hipSetDevice(c->ix->device));
    if (nBytes) HIPCHK(hipMemcpyAsync(text, c->txtOut.as<char>() + offset, nBytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

// test tap: the number format on its own (32 bytes per value, zero-terminated)
extern "C" int kasa_text_dtoa(int device, const double *values, uint32_t n, char *out)
{
    if (!values || !out) return fail(KASA_E_ARG, "kasa_text_dtoa: NULL argument");
    HIPCHK(hipSetDevice(device));
    double *dv = nullptr; char *dout = nullptr;
    if (n == 0) return KASA_OK;
    HIPCHK(hipMalloc(&dv, (size_t)n * 8));
    if (hipMalloc(&dout, (size_t)n * 32) != hipSuccess) { (void)hipFree(dv); return fail(KASA_E_NOMEM, "kasa_text_dtoa: out of device memory"); }
    hipError_t e = hipMemcpy(dv, values, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) { kasa_text::dtoa_probe_kernel<<<blocks_for(n, 256), 256>>>(dv, n, dout); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 32, hipMemcpyDeviceToHost);
    (void)hipFree(dv); (void)hipFree(dout);
    if (e != hipSuccess) return fail(KASA_E_HIP, hipGetErrorString(e));
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// --coherence (Compare::postProcess, Compare.hpp:2607-2728)
// ------------------------------------------------------------------------------------------------
// The reference sorts the batch's k-mers by (read, strand, window) -- the order the encoder emits them in -- and walks
// them ONCE with a little state machine: per read and strand the clusters of overlapping matches (match length = deepest
// matched level of the k-mer, what setMatchLength leaves, Compare.hpp:847-948), score = overlap + 1 - 1 / (times the
// largest overlap occurred), the read keeps the maximum.  The walk carries its read counter itself instead of reading it
// off the elements: a read without k-mers is credited with the first element of its successor, and after a strand switch
// the search for the next match runs on into the following reads.  So where the walk stands when read r's turn begins
// (its ENTRY) depends on the reads before -- but only through that one number: every turn begins with a fresh state.
//
// Here: (1) the reads are encoded once more in emission order and every k-mer's depth is looked up on its own;
// (2) chunks of COH_CHUNK reads are walked in parallel, one thread each, from the entry they would have if the chunk before
// them ended at its last element (the common case); (3) every chunk then compares its entry with the exit its predecessor
// really reached and walks again if they differ, until nothing changes (a round or two: a walk forgets its entry as soon
// as it has seen a match).  The statements of the walk are the reference's, in its order.
static constexpr uint32_t COH_CHUNK = 64;
static constexpr unsigned long long COH_NONE = ~0ull;

template <class Key>
__global__ void coh_depth_kernel(const Key *__restrict__ q, uint64_t nQ, const Key *__restrict__ idxKmer, uint32_t nIdx,
                                 const uint32_t *__restrict__ table, int tb, int kHigh, int kLow, uint8_t *__restrict__ len,
                                 unsigned long long *__restrict__ firstMatch)
{
    const uint64_t o = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (o >= nQ) return;
    const Key k = q[o];
    const uint32_t lo = lower_bound_global<Key>(idxKmer, table, tb, k);
    const int la = (lo < nIdx) ? lcp_letters<Key>(k, idxKmer[lo]) : 0;
    const int lb = (lo > 0) ? lcp_letters<Key>(k, idxKmer[lo - 1]) : 0;
    int L = la >= lb ? la : lb, d = 0;
    if (L >= RANGE_LETTERS) {                                          // as lookup_kernel: the 6-letter prefix exists, '^' ends the query
        if (L > kHigh) L = kHigh;
        d = L;
        for (int kk = kLow; kk <= L; ++kk)
            if (((uint32_t)(k >> (5 * (KeyTraits<Key>::LETTERS - kk))) & 31u) == 30u) { d = kk - 1; break; }
        if (d < kLow) d = 0;
    }
    len[o] = (uint8_t)d;
    if (d) atomicMin(firstMatch, (unsigned long long)o);
}

// where the walk stands among the emitted k-mers: element idx, the read it belongs to and that read's k-mers
struct CohCursor {
    const uint64_t *off; uint32_t nReads; uint32_t strands;
    uint64_t idx, n;
    uint32_t er; uint64_t eStart, eEnd; uint32_t per;                  // read of element idx: its elements eStart .. eEnd - 1, `per` per strand
    __device__ void seek(uint64_t at)
    {
        idx = at;
        if (at >= n) return;
        uint32_t lo = 0, hi = nReads;                                 // last read whose elements start at or before `at`
        while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (off[mid] <= at) lo = mid; else hi = mid; }
        er = lo; eStart = off[er]; eEnd = off[er + 1]; per = (uint32_t)((eEnd - eStart) / strands);
    }
    __device__ void next()
    {
        ++idx;
        while (idx >= eEnd && idx < n) { ++er; eStart = eEnd; eEnd = off[er + 1]; per = (uint32_t)((eEnd - eStart) / strands); }
    }
    __device__ uint32_t read() const { return er; }
    __device__ uint32_t frame() const { return (uint32_t)(idx - eStart) >= per ? 1u : 0u; }
    __device__ uint32_t pos() const { const uint32_t w = (uint32_t)(idx - eStart); return w >= per ? w - per : w; }
};

// turns of the reads r0 .. r1 - 1 from `entry`; the first-match turn (rid == firstRead) starts behind the first match
// with its end remembered (Compare.hpp:2637-2648).  Returns the exit (entry of read r1); fail: the reference's
// vector::at would throw (Compare.hpp:2667 with the index at the end of the batch).
__device__ unsigned long long coh_walk(CohCursor &C, const uint8_t *__restrict__ len, uint32_t r0, uint32_t r1, unsigned long long entry,
                                      uint32_t firstRead, unsigned long long firstIdx, uint32_t firstLast, bool six,
                                      float *__restrict__ scores, bool &fail)
{
    fail = false;
    for (uint32_t r = r0; r < r1; ++r) scores[r] = 0.0f;
    // no match at all, or none before this chunk ends: no turn.  Nothing of it reaches the reads behind (the first-match turn
    // sets its own entry), so the exit is what the next chunk assumed -- returning `entry` made every such chunk's successor
    // walk again, one per round: as many rounds as there are chunks before the first match
    if (firstIdx == COH_NONE || r1 <= firstRead) return C.off[r1];
    uint32_t rid = r0, last = 0xFFFFFFFFu, cur = 0, cnt = 0;
    if (r0 <= firstRead) { rid = firstRead; last = firstLast; entry = firstIdx + 1ull; }
    C.seek(entry);
    auto det = [&](uint32_t nx) { if (nx > cur) { cur = nx; cnt = 1; } else if (nx == cur) cnt++; };              // :2653-2662
    auto cluster = [&]() { const float v = __fsub_rn(__fadd_rn((float)cur, 1.0f), __fdiv_rn(1.0f, (float)cnt)); if (scores[rid] < v) scores[rid] = v; };
    for (; rid < r1 && C.idx < C.n; ++rid) {                           // :2665
        for (uint32_t fb = 0; fb < (six ? 2u : 1u);) {                 // :2667
            if (C.idx >= C.n) { fail = true; return C.n; }
            const uint32_t ml = len[C.idx];
            if (ml != 0u) {
                const uint32_t ps = C.pos();
                if (ps <= last) {
                    if (ps + ml < last) det(ml);
                    else det((uint32_t)((int32_t)last - (int32_t)ps));
                } else { cluster(); cur = 0; }
                last = ps + ml;
            }
            C.next();
            if (C.idx == C.n) { cluster(); break; }
            if (C.read() != rid) { cluster(); last = 0xFFFFFFFFu; cur = 0; cnt = 0; break; }
            if (C.frame() != fb) {
                cluster();
                cur = 0; cnt = 0;
                ++fb;
                while (C.idx < C.n) {
                    const uint32_t m2 = len[C.idx];
                    if (m2 != 0u) { last = C.pos() + m2; C.next(); break; }
                    C.next();
                }
            }
        }
    }
    return C.idx;
}

// round 0: every chunk from its own first element; later rounds: the chunks whose predecessor left somewhere else
__global__ void coh_chunk_kernel(const uint8_t *__restrict__ len, const uint64_t *__restrict__ kmerOff, uint32_t nReads, uint64_t nQ, uint32_t strands,
                                 const unsigned long long *__restrict__ firstMatch, int round, unsigned long long *__restrict__ entryUsed,
                                 unsigned long long *__restrict__ exitIdx, uint32_t *__restrict__ failed, float *__restrict__ scores,
                                 uint32_t *__restrict__ changed)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nChunks = (nReads + COH_CHUNK - 1) / COH_CHUNK;
    if (t >= nChunks) return;
    const uint32_t r0 = t * COH_CHUNK, r1 = min(r0 + COH_CHUNK, nReads);
    unsigned long long entry = kmerOff[r0];
    if (round > 0) {
        if (t == 0) return;
        entry = exitIdx[t - 1];
        if (entry == entryUsed[t]) return;
    }
    CohCursor C;
    C.off = kmerOff; C.nReads = nReads; C.strands = strands; C.n = nQ; C.idx = 0; C.er = 0; C.eStart = 0; C.eEnd = 0; C.per = 0;
    const unsigned long long firstIdx = *firstMatch;
    uint32_t firstRead = 0, firstLast = 0;
    if (firstIdx != COH_NONE) { C.seek(firstIdx); firstRead = C.read(); firstLast = C.pos() + len[firstIdx]; }
    bool fail;
    const unsigned long long ex = coh_walk(C, len, r0, r1, entry, firstRead, firstIdx, firstLast, strands == 2u, scores, fail);
    entryUsed[t] = entry;
    if (round > 0 && (ex != exitIdx[t] || (uint32_t)fail != failed[t])) atomicAdd(changed, 1u);
    else if (round > 0) atomicAdd(changed + 1, 1u);                    // walked again, same exit
    exitIdx[t] = ex;
    failed[t] = fail ? 1u : 0u;
}

__global__ void coh_any_kernel(const uint32_t *__restrict__ failed, uint32_t n, uint32_t *__restrict__ any)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && failed[t]) atomicOr(any, 1u);
}

extern "C" int kasa_batch_coherence(kasa_ctx *c, float *scores, uint64_t *throwsAt)
{
    if (!c || !scores || !throwsAt) return fail(KASA_E_ARG, "kasa_batch_coherence: NULL argument");
    *throwsAt = ~0ull;
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_coherence: batch not sorted");
    if (c->haveSeqRead) return fail(KASA_E_ARG, "kasa_batch_coherence: paired-end input is not supported (the reference's result depends on its unstable sort of the mates' k-mers)");
    if (c->uniqueDone) return fail(KASA_E_ARG, "kasa_batch_coherence: not together with -e (the reference's result depends on which duplicates its unstable sort leaves)");
    if (!c->readsUploaded || c->nSeq != c->nReads) return fail(KASA_E_STATE, "kasa_batch_coherence: the batch was not uploaded as reads");
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t nReads = (uint32_t)c->nReads;
    if (nReads == 0) return KASA_OK;
    // (1) the k-mers once more, in emission order (the sort has consumed them), and the depth of each
    const uint64_t nE = c->nEmitted;
    int rc;
    if ((rc = c->qKmerA.reserve(nE * c->keyBytes() + 64)) || (rc = c->qReadA.reserve(nE * 4 + 64)) || (rc = c->cohLen.reserve(nE + 64))) return rc;
    const uint32_t nChunks = (nReads + COH_CHUNK - 1) / COH_CHUNK;
    if ((rc = c->cohState.reserve((size_t)nChunks * 20 + (size_t)nReads * 4 + 256))) return rc;
    unsigned long long *entryUsed = c->cohState.as<unsigned long long>(), *exitIdx = entryUsed + nChunks;
    uint32_t *failed = reinterpret_cast<uint32_t *>(exitIdx + nChunks);
    float *dScores = reinterpret_cast<float *>(failed + nChunks + (nChunks & 1u));
    unsigned long long *firstMatch = c->misc.as<unsigned long long>() + 26;
    uint32_t *changed = c->misc.as<uint32_t>() + 54;                                   // [54] changed exits, [55] same exits, [56] any failure
    HIPCHK(hipMemsetAsync(firstMatch, 0xFF, 8, c->stream));
    HIPCHK(hipMemsetAsync(changed, 0, 12, c->stream));
    if (nE > 0) {
        const unsigned blocks = (unsigned)std::min<int64_t>((c->nSeq + ENC_WAVES - 1) / ENC_WAVES, 256 * 16);
        if (c->ix->wide) {
            encode_kernel<key128><<<blocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(), c->seqOff.as<uint64_t>(), nullptr, c->nSeq,
                c->kLow, c->strands(), c->enc_mode(), c->lut.as<uint8_t>(), c->qKmerA.as<key128>(), c->qReadA.as<uint32_t>(), 0, nullptr, nullptr, ~0ull);
            coh_depth_kernel<key128><<<blocks_for(nE, 256), 256, 0, c->stream>>>(c->qKmerA.as<key128>(), nE, c->ix->kmer.as<key128>(), (uint32_t)c->ix->n,
                c->ix->table.as<uint32_t>(), c->ix->tb, c->kHigh, c->kLow, c->cohLen.as<uint8_t>(), firstMatch);
        } else {
            encode_kernel<uint64_t><<<blocks, 64 * ENC_WAVES, 0, c->stream>>>(c->basesPtr, c->baseOff.as<int64_t>(), c->seqOff.as<uint64_t>(), nullptr, c->nSeq,
                c->kLow, c->strands(), c->enc_mode(), c->lut.as<uint8_t>(), c->qKmerA.as<uint64_t>(), c->qReadA.as<uint32_t>(), 0, nullptr, nullptr, ~0ull);
            coh_depth_kernel<uint64_t><<<blocks_for(nE, 256), 256, 0, c->stream>>>(c->qKmerA.as<uint64_t>(), nE, c->ix->kmer.as<uint64_t>(), (uint32_t)c->ix->n,
                c->ix->table.as<uint32_t>(), c->ix->tb, c->kHigh, c->kLow, c->cohLen.as<uint8_t>(), firstMatch);
        }
        HIPCHK(hipGetLastError());
    }
    // (2), (3) the walk: all chunks, then the chunks whose entry was not what their predecessor left, until none is
    const uint64_t *emitOff = c->seqOff.as<uint64_t>();                                // k-mers before every read, as emitted (kmerOff follows -e)
    for (int round = 0;; ++round) {
        coh_chunk_kernel<<<blocks_for(nChunks, 64), 64, 0, c->stream>>>(c->cohLen.as<uint8_t>(), emitOff, nReads, nE, (uint32_t)c->strands(), firstMatch, round,
                                                                        entryUsed, exitIdx, failed, dScores, changed);
        HIPCHK(hipGetLastError());
        if (round == 0) continue;                                                      // (round 1 compares with round 0's exits)
        uint32_t h[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h, changed, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemsetAsync(changed, 0, 8, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (h[0] == 0u) break;                                                         // every recomputed chunk left where it had left before
        if (round > (int)nChunks + 2) return fail(KASA_E_LIMIT, "kasa_batch_coherence: the walk did not settle");
    }
    coh_any_kernel<<<blocks_for(nChunks, 256), 256, 0, c->stream>>>(failed, nChunks, changed + 2);
    uint32_t anyFail = 0;
    HIPCHK(hipMemcpyAsync(&anyFail, changed + 2, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(scores, dScores, (size_t)nReads * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (anyFail) *throwsAt = nE;
    c->cohScores = dScores;
    return KASA_OK;
}

// Page-locked host memory for the buffers that cross PCIe (reads in, ranked hits or CSR out): transfers from pageable
// memory are staged by the runtime at a fraction of the link rate.
extern "C" void *kasa_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void kasa_host_free(void *p) { if (p) (void)hipHostFree(p); }
// Plain device buffers for a host that keeps its inputs resident (kasa_batch_upload_device) without another GPU library in
// the process: bench.py's one-GPU run allocates its reads through these and never imports torch, so the library runs on the
// HIP runtime it was built for.
extern "C" int kasa_device_alloc(int device, size_t bytes, void **out)
{
    if (!out) return fail(KASA_E_ARG, "kasa_device_alloc: NULL argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    if (hipMalloc(out, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return fail(KASA_E_NOMEM, "kasa_device_alloc: %zu bytes", bytes); }
    return KASA_OK;
}
extern "C" int kasa_device_free(int device, void *p)
{
    if (!p) return KASA_OK;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(p));
    return KASA_OK;
}
extern "C" int kasa_device_read(int device, void *dst, const void *src, size_t bytes)
{
    if (bytes && (!dst || !src)) return fail(KASA_E_ARG, "kasa_device_read: NULL argument");
    HIPCHK(hipSetDevice(device));
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return KASA_OK;
}
extern "C" int kasa_device_write(int device, void *dst, const void *src, size_t bytes)
{
    if (bytes && (!dst || !src)) return fail(KASA_E_ARG, "kasa_device_write: NULL argument");
    HIPCHK(hipSetDevice(device));
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return KASA_OK;
}
extern "C" int kasa_thread_device(int device)
{
    HIPCHK(hipSetDevice(device));
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// profile tables
// ------------------------------------------------------------------------------------------------
// the three accumulators of a cell folded into a 128-bit value {hi, lo}
static int fetch_tables(kasa_ctx *c, std::vector<uint64_t> &u, std::vector<uint64_t> &t, std::vector<uint64_t> &hi, std::vector<uint64_t> &lo)
{
    HIPCHK(hipSetDevice(c->ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t cells = (size_t)c->nK * c->ix->nTaxa;
    u.resize(cells); t.resize(cells); hi.resize(cells); lo.resize(cells);
    HIPCHK(hipMemcpy(u.data(), c->cntUnique.p, cells * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(t.data(), c->cntTotal.p, cells * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hi.data(), c->cntAllHi.p, cells * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lo.data(), c->cntAllLo.p, cells * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> mid(cells);
    HIPCHK(hipMemcpy(mid.data(), c->cntAllMid.p, cells * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cells; ++i) {
        const unsigned __int128 v = ((unsigned __int128)hi[i] << 64) + ((unsigned __int128)mid[i] << 32) + lo[i];
        hi[i] = (uint64_t)(v >> 64); lo[i] = (uint64_t)v;
    }
    return KASA_OK;
}

static int profile_fetch_impl(kasa_ctx *c, double *countAll, uint64_t *countUnique, uint64_t *countTotal)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    std::vector<uint64_t> u, t, hi, lo;
    int rc = fetch_tables(c, u, t, hi, lo);
    if (rc) return rc;
    for (size_t i = 0; i < u.size(); ++i) {
        if (countUnique) countUnique[i] = u[i];
        if (countTotal) countTotal[i] = t[i];
        if (countAll) countAll[i] = (double)hi[i] + (double)lo[i] * 5.42101086242752217e-20; // 2^-64
    }
    return KASA_OK;
}

__global__ void tables_absorb_kernel(uint64_t *__restrict__ dst, uint64_t *__restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { dst[i] += src[i]; src[i] = 0ull; }
}

// dst += src, src = 0: the tables of a context that only grouped (a partition worker: the profile is made where the
// queries are grouped) go to the context that owns the file's profile.  Same device, same k range, same index content.
extern "C" int kasa_profile_absorb(kasa_ctx *dst, kasa_ctx *src)
{
    if (!dst || !src) return fail(KASA_E_ARG, "kasa_profile_absorb: NULL argument");
    if (dst == src) return KASA_OK;
    if (dst->device != src->device || dst->nK != src->nK || dst->kHigh != src->kHigh || dst->ix->nTaxa != src->ix->nTaxa)
        return fail(KASA_E_ARG, "kasa_profile_absorb: the contexts differ in device, k range or number of taxa");
    HIPCHK(hipSetDevice(dst->device));
    HIPCHK(hipStreamSynchronize(src->stream));
    const size_t cells = (size_t)dst->nK * dst->ix->nTaxa;
    DevBuf *d[5] = {&dst->cntUnique, &dst->cntTotal, &dst->cntAllHi, &dst->cntAllMid, &dst->cntAllLo};
    DevBuf *q[5] = {&src->cntUnique, &src->cntTotal, &src->cntAllHi, &src->cntAllMid, &src->cntAllLo};
    for (int k = 0; k < 5; ++k)
        tables_absorb_kernel<<<blocks_for(cells, 256), 256, 0, dst->stream>>>(d[k]->as<uint64_t>(), q[k]->as<uint64_t>(), cells);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(dst->stream));
    return KASA_OK;
}

extern "C" int kasa_profile_fetch(kasa_ctx *c, double *countAll, uint64_t *countUnique, uint64_t *countTotal)
{
    KASA_GUARDED(profile_fetch_impl(c, countAll, countUnique, countTotal))
}

static int profile_export_limbs_impl(kasa_ctx *c, uint64_t *limbs)
{
    if (!c || !limbs) return fail(KASA_E_ARG, "kasa_profile_export_limbs: NULL argument");
    std::vector<uint64_t> u, t, hi, lo;
    int rc = fetch_tables(c, u, t, hi, lo);
    if (rc) return rc;
    for (size_t i = 0; i < u.size(); ++i) {
        uint64_t *o = limbs + i * 6;
        o[0] = u[i]; o[1] = t[i];
        o[2] = lo[i] & 0xFFFFFFFFull; o[3] = lo[i] >> 32; o[4] = hi[i] & 0xFFFFFFFFull; o[5] = hi[i] >> 32;
    }
    return KASA_OK;
}

extern "C" int kasa_profile_export_limbs(kasa_ctx *c, uint64_t *limbs)
{
    KASA_GUARDED(profile_export_limbs_impl(c, limbs))
}

static int profile_import_limbs_impl(kasa_ctx *c, const uint64_t *limbs)
{
    if (!c || !limbs) return fail(KASA_E_ARG, "kasa_profile_import_limbs: NULL argument");
    HIPCHK(hipSetDevice(c->ix->device));
    const size_t cells = (size_t)c->nK * c->ix->nTaxa;
    std::vector<uint64_t> u(cells), t(cells), hi(cells), lo(cells);
    for (size_t i = 0; i < cells; ++i) {
        const uint64_t *o = limbs + i * 6;
        u[i] = o[0]; t[i] = o[1];
        unsigned __int128 v = (unsigned __int128)o[2] + ((unsigned __int128)o[3] << 32) + ((unsigned __int128)o[4] << 64) + ((unsigned __int128)o[5] << 96);
        lo[i] = (uint64_t)v; hi[i] = (uint64_t)(v >> 64);
    }
    HIPCHK(hipMemcpy(c->cntUnique.p, u.data(), cells * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->cntTotal.p, t.data(), cells * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->cntAllHi.p, hi.data(), cells * 8, hipMemcpyHostToDevice));
    std::vector<uint64_t> mid(cells);
    for (size_t i = 0; i < cells; ++i) { mid[i] = lo[i] >> 32; lo[i] &= 0xFFFFFFFFull; }
    HIPCHK(hipMemcpy(c->cntAllLo.p, lo.data(), cells * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->cntAllMid.p, mid.data(), cells * 8, hipMemcpyHostToDevice));
    return KASA_OK;
}

extern "C" int kasa_profile_import_limbs(kasa_ctx *c, const uint64_t *limbs)
{
    KASA_GUARDED(profile_import_limbs_impl(c, limbs))
}

// ---- the multi-GPU reduce of the profile tables (Compare.hpp:3445-3454 across devices) -----------------------------
// Every cell leaves as six u64 limbs {unique, total, all[0..3] (32 bits each)}; integer sums of limbs are exact and
// independent of the order in which ranks arrive, so one ncclAllReduce(u64, sum) over xGMI gives every rank the same
// global tables; the carries are folded back on the device.
__global__ void limbs_pack_kernel(const uint64_t *__restrict__ u, const uint64_t *__restrict__ t, const uint64_t *__restrict__ hi,
                                  const uint64_t *__restrict__ mid, const uint64_t *__restrict__ lo, size_t cells, uint64_t *__restrict__ limbs)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const unsigned __int128 v = ((unsigned __int128)hi[i] << 64) + ((unsigned __int128)mid[i] << 32) + lo[i];
    uint64_t *o = limbs + i * 6;
    o[0] = u[i]; o[1] = t[i];
    o[2] = (uint64_t)v & 0xFFFFFFFFull; o[3] = (uint64_t)(v >> 32) & 0xFFFFFFFFull;
    o[4] = (uint64_t)(v >> 64) & 0xFFFFFFFFull; o[5] = (uint64_t)(v >> 96);
}

__global__ void limbs_unpack_kernel(const uint64_t *__restrict__ limbs, size_t cells, uint64_t *__restrict__ u, uint64_t *__restrict__ t,
                                    uint64_t *__restrict__ hi, uint64_t *__restrict__ mid, uint64_t *__restrict__ lo)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const uint64_t *o = limbs + i * 6;
    const unsigned __int128 v = (unsigned __int128)o[2] + ((unsigned __int128)o[3] << 32) + ((unsigned __int128)o[4] << 64) + ((unsigned __int128)o[5] << 96);
    u[i] = o[0]; t[i] = o[1];
    hi[i] = (uint64_t)(v >> 64); mid[i] = (uint64_t)v >> 32; lo[i] = (uint64_t)v & 0xFFFFFFFFull;
}

// RCCL is bound at run time, to the copy the PROCESS already has: the communicator the caller hands over was made by that
// copy (the host's own link, torch's bundled librccl, a ctypes load), and a second RCCL -- or ROCm's RCCL over another HIP
// runtime than the one it was built for -- has no business in the process.  Only when none is loaded: librccl.so.1 by the
// usual search (this library's RUNPATH: /opt/rocm/lib).
struct RcclApi {
    ncclResult_t (*allReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*errorString)(ncclResult_t) = nullptr;
    ncclResult_t (*getVersion)(int *) = nullptr;
    bool loadedHere = false;                            // no copy was in the process: this library loaded one
};
static const RcclApi *rccl_api(bool mayLoad)
{
    static std::mutex mu;
    static RcclApi api;
    std::lock_guard<std::mutex> lock(mu);
    if (api.allReduce) return &api;
    void *h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = RTLD_DEFAULT;                            // linked into the host, or loaded globally
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);                             // loaded privately (ctypes, torch): found by its SONAME
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h && mayLoad) { h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL); api.loadedHere = h != nullptr; }
    if (!h) return nullptr;
    api.errorString = reinterpret_cast<decltype(api.errorString)>(dlsym(h, "ncclGetErrorString"));
    api.getVersion = reinterpret_cast<decltype(api.getVersion)>(dlsym(h, "ncclGetVersion"));
    api.allReduce = reinterpret_cast<decltype(api.allReduce)>(dlsym(h, "ncclAllReduce"));
    return api.allReduce ? &api : nullptr;
}

// What this library was built with and what it runs on (HIP: major * 10^7 + minor * 10^5 + patch; RCCL: its own code, 0 when
// no RCCL is in the process).  A host that shares its process with another ROCm stack (a torch wheel brings its own
// libamdhip64 / librccl) can see -- and refuse -- a mismatch.
extern "C" int kasa_runtime_versions(int *hipBuilt, int *hipRuntime, int *hipDriver, int *rcclBuilt, int *rcclRuntime)
{
    if (hipBuilt) *hipBuilt = HIP_VERSION;
    if (hipRuntime) { int v = 0; if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); v = 0; } *hipRuntime = v; }
    if (hipDriver) { int v = 0; if (hipDriverGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); v = 0; } *hipDriver = v; }
    if (rcclBuilt) *rcclBuilt = NCCL_VERSION_CODE;
    if (rcclRuntime) {
        int v = 0;
        const RcclApi *r = rccl_api(false);
        if (r && r->getVersion) (void)r->getVersion(&v);
        *rcclRuntime = v;
    }
    return KASA_OK;
}

extern "C" int kasa_profile_allreduce(kasa_ctx *c, void *rcclComm)
{
    if (!c || !rcclComm) return fail(KASA_E_ARG, "kasa_profile_allreduce: NULL argument");
    const RcclApi *R = rccl_api(true);
    if (!R) return fail(KASA_E_HIP, "kasa_profile_allreduce: no RCCL library in this process and librccl.so.1 cannot be loaded: %s", dlerror());
    HIPCHK(hipSetDevice(c->ix->device));
    const size_t cells = (size_t)c->nK * c->ix->nTaxa;
    int rc = c->profSorted.reserve(cells * 6 * 8 + 64);                 // (free between batches)
    if (rc) return rc;
    uint64_t *limbs = c->profSorted.as<uint64_t>();
    limbs_pack_kernel<<<blocks_for(cells, 256), 256, 0, c->stream>>>(c->cntUnique.as<uint64_t>(), c->cntTotal.as<uint64_t>(), c->cntAllHi.as<uint64_t>(),
        c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>(), cells, limbs);
    HIPCHK(hipGetLastError());
    const ncclResult_t nr = R->allReduce(limbs, limbs, cells * 6, ncclUint64, ncclSum, static_cast<ncclComm_t>(rcclComm), c->stream);
    if (nr != ncclSuccess) return fail(KASA_E_HIP, "ncclAllReduce failed: %s", R->errorString ? R->errorString(nr) : "?");
    limbs_unpack_kernel<<<blocks_for(cells, 256), 256, 0, c->stream>>>(limbs, cells, c->cntUnique.as<uint64_t>(), c->cntTotal.as<uint64_t>(),
        c->cntAllHi.as<uint64_t>(), c->cntAllMid.as<uint64_t>(), c->cntAllLo.as<uint64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

// ------------------------------------------------------------------------------------------------
// measurement + test taps
// ------------------------------------------------------------------------------------------------
extern "C" int kasa_ctx_stage_ms(kasa_ctx *c, int stage, double *ms, uint64_t *launches)
{
    if (!c || stage < 0 || stage >= KASA_STAGE_COUNT) return fail(KASA_E_ARG, "kasa_ctx_stage_ms: bad argument");
    HIPCHK(hipSetDevice(c->ix->device));
    int rc = timer_resolve(c->timers[stage]);
    if (rc) return rc;
    if (ms) *ms = c->timers[stage].ms;
    if (launches) *launches = c->timers[stage].launches;
    return KASA_OK;
}

extern "C" int kasa_ctx_kernel_ms(kasa_ctx *c, int kernel, double *ms, uint64_t *launches)
{
    if (!c || kernel < 0 || kernel >= KASA_KERNEL_COUNT) return fail(KASA_E_ARG, "kasa_ctx_kernel_ms: bad argument");
    HIPCHK(hipSetDevice(c->ix->device));
    int rc = timer_resolve(c->kernels[kernel]);
    if (rc) return rc;
    if (ms) *ms = c->kernels[kernel].ms;
    if (launches) *launches = c->kernels[kernel].launches;
    return KASA_OK;
}

extern "C" int kasa_ctx_batch_stats(kasa_ctx *c, uint64_t *stats)
{
    if (!c || !stats) return fail(KASA_E_ARG, "kasa_ctx_batch_stats: NULL argument");
    stats[0] = c->nQ; stats[1] = c->lastStaged; stats[2] = c->lastKeys; stats[3] = c->poolUsed;
    stats[4] = c->lastSlowReads; stats[5] = c->lastOverflowReads; stats[6] = c->nnz; stats[7] = c->payloadIsSlot ? 1 : 0;
    return KASA_OK;
}


// ---- diagnostic (not part of the ABI): shape of the records score_other_kernel sees, RW = 8 only
__global__ __launch_bounds__(256) void record_stats_kernel(ScoreArgs A, unsigned long long *out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const bool inRange = slot < A.nQ;
    uint4 cur[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    uint4 mo = make_uint4(0, 0, 0, 0);
    if (inRange) {
        cur[0] = reinterpret_cast<const uint4 *>(A.rec)[(size_t)slot * (A.recCW / 4u)]; cur[1] = reinterpret_cast<const uint4 *>(A.rec)[(size_t)slot * (A.recCW / 4u) + 1];
        uint32_t lo = 0, hi = A.nReads;
        while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (A.kmerOff[mid] <= slot) lo = mid; else hi = mid; }
        mo = reinterpret_cast<const uint4 *>(A.mainOut)[lo];
    }
    const bool live = inRange && mo.w != 0u && (cur[0].z & 31u) != 0u;
    QueryRec<8> Q;
    Q.decode_regs(cur, A.pool);
    if (!live) { Q.d = 0; Q.nInl = 0; Q.nMore = 0; Q.split = 0; Q.nseg = 0; }
    const bool sat = (cur[0].z & REC_SAT) != 0u, split = live && (cur[0].z & REC_SPLIT) != 0u;
    const int nEv = Q.d ? Q.d - A.kLow + 1 : 0;
    unsigned long long v[32];
    for (int i = 0; i < 32; ++i) v[i] = 0;
    v[0] = live; v[1] = live && Q.nMore > 0; v[2] = split; v[3] = live && sat; v[4] = Q.nseg;
    uint32_t nOtherSeg = 0, rec = 0;
    for (uint32_t q = 0; q < Q.nseg; ++q) {
        const uint32_t sq = q < Q.nInl ? Q.sg[q] : Q.more[q - Q.nInl];
        const uint32_t t = sq & SEG_TAX_MASK, m = seg_level_mask(sq, A.kHigh);
        const uint32_t pc = __popc(m);
        if (t != mo.x && t != mo.y) {
            ++nOtherSeg; v[5]++; v[6] += pc == 1; v[7] += pc == 2; v[8] += pc >= 3;
            if (!split && !sat) v[16] += pc >= 3;
            rec += seg_records<8>(m, split);
        }
    }
    v[18] = rec; v[10] = split ? rec : 0; v[11] = (live && sat) ? rec : 0; v[17] = live && rec == 0; v[19] = Q.nMore;
    // order monotone?
    bool asc = true, desc = true;
    { uint32_t o = (uint32_t)Q.order; int prev = -1; for (int ev = 0; ev < nEv; ++ev, o >>= 3) { const int lv = o & 7; if (prev >= 0) { if (lv < prev) asc = false; if (lv > prev) desc = false; } prev = lv; } }
    v[14] = live && (asc || desc);
    v[20] = live && nOtherSeg == 0; v[21] = live && nOtherSeg == 1; v[22] = live && nOtherSeg == 2; v[23] = live && nOtherSeg >= 3;
    v[24] = live && Q.nseg == 1; v[25] = live && Q.nseg == 2; v[26] = live && Q.nseg == 3; v[27] = live && Q.nseg == 4; v[28] = live && Q.nseg > 4 && Q.nseg <= 8; v[29] = live && Q.nseg > 8;
    // wavefront maxima
    uint32_t mx = Q.nMore, ms = Q.nseg, mo2 = nOtherSeg;
    for (int off = 32; off; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); ms = max(ms, (uint32_t)__shfl_xor((int)ms, off)); mo2 = max(mo2, (uint32_t)__shfl_xor((int)mo2, off)); }
    v[12] = lane == 0 ? mx : 0; v[13] = lane == 0 ? ms : 0; v[15] = lane == 0; v[30] = lane == 0 ? mo2 : 0;
    v[31] = lane == 0 && __ballot(split) != 0ull;
    for (int i = 0; i < 32; ++i) {
        unsigned long long x = v[i];
        for (int off = 32; off; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0 && x) atomicAdd(&out[i], x);
    }
}

extern "C" int kasa_debug_record_stats(kasa_ctx *c, uint64_t *out32)
{
    if (!c || !out32) return fail(KASA_E_ARG, "kasa_debug_record_stats: NULL argument");
    if (c->state < 4 || !c->grouped || c->recWords() != 8) return fail(KASA_E_STATE, "kasa_debug_record_stats: needs a scored batch with 32-byte records (before kasa_batch_rank)");
    HIPCHK(hipSetDevice(c->ix->device));
    ScopedBuf tmp;
    int rc = tmp.reserve(32 * 8);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(tmp.p, 0, 32 * 8, c->stream));
    ScoreArgs A;
    memset(&A, 0, sizeof(A));
    A.rec = c->rec.as<uint32_t>(); A.recCW = c->recCW; A.recQS = c->recCW == 8u ? 1u : 2u; A.rowPerQuery = 0u; A.kmerOff = c->kmerOff.as<uint64_t>(); A.pool = c->pool.as<uint32_t>();
    A.nReads = (uint32_t)c->nReads; A.kHigh = c->kHigh; A.kLow = c->kLow; A.nQ = (uint32_t)c->nQ; A.mainOut = c->fastScratch.as<uint32_t>();
    record_stats_kernel<<<blocks_for(c->nQ, 256), 256, 0, c->stream>>>(A, tmp.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out32, tmp.p, 32 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

extern "C" int kasa_ctx_stage_reset(kasa_ctx *c)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->ix->device));
    for (auto &t : c->timers) { int rc = timer_resolve(t); if (rc) return rc; t.ms = 0; t.launches = 0; }
    for (auto &t : c->kernels) { int rc = timer_resolve(t); if (rc) return rc; t.ms = 0; t.launches = 0; }
    return KASA_OK;
}

extern "C" int kasa_batch_query_count(kasa_ctx *c, uint64_t *n)
{
    if (!c || !n) return fail(KASA_E_ARG, "kasa_batch_query_count: NULL argument");
    if (c->state < 1) return fail(KASA_E_STATE, "kasa_batch_query_count: no batch uploaded");
    *n = c->nQ;
    return KASA_OK;
}

extern "C" int kasa_batch_fetch_queries(kasa_ctx *c, void *kmers, uint32_t *reads, uint64_t n)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (c->state < 2) return fail(KASA_E_STATE, "kasa_batch_fetch_queries: batch not encoded");
    if (n > c->nQ) return fail(KASA_E_ARG, "kasa_batch_fetch_queries: n exceeds the batch");
    HIPCHK(hipSetDevice(c->ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n && kmers) HIPCHK(hipMemcpy(kmers, c->qKmer, n * c->keyBytes(), hipMemcpyDeviceToHost));
    if (n && reads) {
        const uint32_t *src = c->qRead;
        if (c->payloadIsSlot) {                                       // the payload is the slot: back to read ids for the caller
            int rc = c->plist.reserve(n * 4 + 64);
            if (rc) return rc;
            slot_to_read_kernel<<<blocks_for(n, 256), 256, 0, c->stream>>>(c->qRead, (uint32_t)n, c->kmerOff.as<uint64_t>(), (uint32_t)c->nReads, c->plist.as<uint32_t>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));
            src = c->plist.as<uint32_t>();
        }
        HIPCHK(hipMemcpy(reads, src, n * 4, hipMemcpyDeviceToHost));
    }
    return KASA_OK;
}

static int batch_set_queries_impl(kasa_ctx *c, const void *kmers, const uint32_t *reads, uint64_t n, int64_t nReads)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (nReads < 0 || (n && (!kmers || !reads))) return fail(KASA_E_ARG, "kasa_batch_set_queries: bad arguments");
    if (n >= 0xFFFFFFF0ull) return fail(KASA_E_LIMIT, "kasa_batch_set_queries: too many queries for one batch");
    HIPCHK(hipSetDevice(c->ix->device));
    c->state = 0; c->haveScores = false; c->grouped = false; c->slotOf = nullptr; c->payloadIsSlot = false; c->rankValid = false; c->txtValid = false; c->cohScores = nullptr; c->readsUploaded = false; c->recOut = nullptr; c->recSorted = false;
    std::vector<uint64_t> koff((size_t)nReads + 1, 0);
    uint32_t maxCnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if ((int64_t)reads[i] >= nReads) return fail(KASA_E_ARG, "kasa_batch_set_queries: read id out of range");
        const bool tooWide = c->ix->wide ? ((static_cast<const uint64_t *>(kmers)[2 * i + 1] >> 61) != 0)     // high word of {lo, hi}
                                         : ((static_cast<const uint64_t *>(kmers)[i] >> 60) != 0);
        if (tooWide) return fail(KASA_E_ARG, "kasa_batch_set_queries: k-mer uses more than %d bits", c->ix->wide ? 125 : 60);
        koff[(size_t)reads[i] + 1]++;
    }
    for (int64_t r = 0; r < nReads; ++r) { maxCnt = std::max<uint32_t>(maxCnt, (uint32_t)koff[(size_t)r + 1]); koff[(size_t)r + 1] += koff[(size_t)r]; }
    int rc;
    if ((rc = c->qKmerA.reserve(n * c->keyBytes() + 64)) || (rc = c->qReadA.reserve(n * 4 + 64)) || (rc = c->kmerOff.reserve(((size_t)nReads + 1) * 8)))
        return rc;
    if (n) HIPCHK(hipMemcpyAsync(c->qKmerA.p, kmers, n * c->keyBytes(), hipMemcpyHostToDevice, c->stream));
    if (n) HIPCHK(hipMemcpyAsync(c->qReadA.p, reads, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->kmerOff.p, koff.data(), ((size_t)nReads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->nReads = nReads; c->nQ = n; c->maxCnt = maxCnt;
    c->qKmer = c->qKmerA.p; c->qRead = c->qReadA.as<uint32_t>();
    c->state = 2;
    return KASA_OK;
}

extern "C" int kasa_batch_set_queries(kasa_ctx *c, const void *kmers, const uint32_t *reads, uint64_t n, int64_t nReads)
{
    KASA_GUARDED(batch_set_queries_impl(c, kmers, reads, n, nReads))
}

extern "C" int kasa_batch_fetch_lookup(kasa_ctx *c, uint8_t *depth, uint32_t *indexPos, uint64_t n)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (c->state < 3) return fail(KASA_E_STATE, "kasa_batch_fetch_lookup: batch not sorted");
    if (n > c->nQ) return fail(KASA_E_ARG, "kasa_batch_fetch_lookup: n exceeds the batch");
    HIPCHK(hipSetDevice(c->ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n && depth) HIPCHK(hipMemcpy(depth, c->depth.p, n, hipMemcpyDeviceToHost));
    if (n && indexPos) HIPCHK(hipMemcpy(indexPos, c->rep.p, n * 4, hipMemcpyDeviceToHost));
    return KASA_OK;
}

extern "C" int kasa_ctx_device_bytes(kasa_ctx *c, uint64_t *bytes)
{
    if (!c || !bytes) return fail(KASA_E_ARG, "NULL argument");
    const std::vector<DevBuf *> all = c->buffers();
    uint64_t s = 0;
    for (const DevBuf *b : all) s += b->cap;
    *bytes = s;
    return KASA_OK;
}

extern "C" int kasa_ctx_debug(kasa_ctx *c, int forceSlowScore, uint32_t *lastSlowReads)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (forceSlowScore >= 0) { c->forceSlowScore = (forceSlowScore & 1) != 0; c->lookupMode = (forceSlowScore & 2) ? 1 : 0; c->debugFlags = forceSlowScore; }
    if (lastSlowReads) *lastSlowReads = c->lastSlowReads;
    if (getenv("KASA_DEBUG_WHY")) {
        uint32_t w[8];
        if (hipMemcpy(w, c->misc.as<uint32_t>() + 8, 32, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[kasa] fallback reasons: cnt=%u taxa=%u log=%u big=%u pending=%u; general kernel: %u reads, %u of them handed to the second pass\n",
                    w[0], w[1], w[2], w[3], w[4], c->lastSlowReads, c->lastOverflowReads);
    }
    return KASA_OK;
}

extern "C" int kasa_device_memory(int device, uint64_t *freeBytes, uint64_t *totalBytes)
{
    int ndev = 0;
    kasa_device_count(&ndev);
    if (device < 0 || device >= ndev) return fail(KASA_E_HIP, "kasa_device_memory: no HIP device %d (found %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (freeBytes) *freeBytes = f;
    if (totalBytes) *totalBytes = t;
    return KASA_OK;
}

// Device bytes one query (k-mer of a read) costs while its batch is in flight: both sort buffers, depth + index
// position, its event record, slot + position list (slot fix-up), and a share for pool, staging rows and the CSR.
extern "C" uint64_t kasa_batch_bytes_per_query(const kasa_ctx *c)
{
    if (!c) return 0;
    return 2 * (c->keyBytes() + 4) + 5 + 4ull * (uint64_t)c->recWords() + 8 + 40;
}

// Room for a batch of about nQueries k-mers out of nBases bases before it arrives: hipMalloc takes 25-90 ms per GB on this
// platform (the 44 GB of a 10 M-read batch's event records: 1.2-4 s), which a host can spend while it still parses the input.
// Only sizes: what a batch needs beyond this is allocated when it comes, as always.  Not while a batch is in flight on ctx.
extern "C" int kasa_ctx_reserve(kasa_ctx *c, uint64_t nQueries, uint64_t nBases, int wantPerRead)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->ix->device));
    const size_t nQ = (size_t)nQueries;
    int rc;
    if ((rc = c->bases.reserve((size_t)nBases + 64)) || (rc = c->qKmerA.reserve(nQ * c->keyBytes() + 64)) || (rc = c->qReadA.reserve(nQ * 4 + 64)) ||
        (rc = c->qKmerB.reserve(nQ * c->keyBytes() + 64)) || (rc = c->qReadB.reserve(nQ * 4 + 64)) || (rc = c->depth.reserve(nQ + 64)) || (rc = c->rep.reserve(nQ * 4 + 64)) ||
        (rc = c->recWords() == 8u ? reserve_placed(c->rec, nQ * (size_t)32 + 64, c->stream, &c->recPlacement)
                                  : c->rec.reserve(nQ * (size_t)c->recWords() * 4 + 64)))   // (64-byte cells for narrow records, where asked for: group_stage)
        return rc;
    (void)wantPerRead;
    return KASA_OK;
}

extern "C" int kasa_ctx_dense_reads(kasa_ctx *c, uint32_t *denseReads)
{
    if (!c || !denseReads) return fail(KASA_E_ARG, "kasa_ctx_dense_reads: NULL argument");
    *denseReads = c->lastDenseReads;
    return KASA_OK;
}

extern "C" int kasa_ctx_group_tiles(kasa_ctx *c, uint32_t *tiles, uint32_t *listed)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (tiles) *tiles = (uint32_t)((c->nQ + TILE - 1) / TILE);
    if (listed) *listed = c->lastSlowTiles;
    return KASA_OK;
}

extern "C" int kasa_ctx_replay_stats(kasa_ctx *c, uint32_t *reads, uint64_t *events)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (reads) *reads = c->lastReplayReads;
    if (events) *events = c->lastReplayEvents;
    return KASA_OK;
}

extern "C" int kasa_ctx_record_placement(kasa_ctx *c, uint32_t *candidates, float *keptRate, float *rates4)
{
    if (!c || !candidates || !keptRate) return fail(KASA_E_ARG, "kasa_ctx_record_placement: NULL argument");
    *candidates = c->recPlacement.candidates;
    *keptRate = c->recPlacement.kept;
    if (rates4) for (int i = 0; i < 4; ++i) rates4[i] = c->recPlacement.rates[i];
    return KASA_OK;
}

extern "C" int kasa_ctx_group_second_chance(kasa_ctx *c, uint32_t *listedAgain)
{
    if (!c || !listedAgain) return fail(KASA_E_ARG, "kasa_ctx_group_second_chance: NULL argument");
    *listedAgain = c->lastSlowTiles2;
    return KASA_OK;
}

extern "C" int kasa_ctx_third_pass_reads(kasa_ctx *c, uint32_t *thirdPassReads)
{
    if (!c || !thirdPassReads) return fail(KASA_E_ARG, "kasa_ctx_third_pass_reads: NULL argument");
    *thirdPassReads = c->lastThirdPassReads;
    return KASA_OK;
}

extern "C" int kasa_ctx_counters(kasa_ctx *c, uint32_t *generalReads, uint32_t *secondPassReads)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    if (generalReads) *generalReads = c->lastSlowReads;
    if (secondPassReads) *secondPassReads = c->lastOverflowReads;
    return KASA_OK;
}

extern "C" int kasa_ctx_synchronize(kasa_ctx *c)
{
    if (!c) return fail(KASA_E_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return KASA_OK;
}

// kasa_radix.h -- the query sort's radix passes, hand-written for gfx950 (wave64, 160 KiB LDS per CU).
//
// Replaces the reference's sort of the batch's k-mers (source/utils/ParallelQuicksort.hpp, Compare.hpp:1077,1123-1132).
// A stable LSD radix sort of (key, 32-bit payload) pairs over a window of key bits, 8 bits per pass, one kernel launch per
// pass ("onesweep": the digit counts of ALL passes come from one read of the keys; a pass finds where a tile's digits go by
// looking back over the tiles before it while those are still running).  What differs from the library's passes:
//   * a tile is 8192 pairs of 8-byte keys (4096 of 16-byte keys), one 1024-thread workgroup: a digit's run inside a tile is
//     32 pairs on average = 256 contiguous bytes of keys, the size from which scattered stores run at streaming rate on this
//     chip (MI355X_MICROARCH.md, "plain stores ... 256 B per wave-instruction"); the library's tiles give runs half as long;
//   * keys and payloads pass through the SAME LDS area one after the other (64 KiB + tables), so two such workgroups are
//     resident per CU: one loads and ranks while the other drains;
//   * ranking is by wave-wide digit matching (ballots), counts kept per wave in LDS: no atomics, stable by construction;
//   * input and output ping-pong between the caller's two buffers: no third copy of the batch.
// Inter-workgroup protocol (look-back): one 64-bit word per (tile, digit) = {2-bit state, 62-bit count}, written and polled
// with agent-scope atomics (the per-XCD L2s are not coherent); tiles are numbered in the order the workgroups START, so a
// tile only ever waits for tiles that are already running.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace kasa_radix {

static constexpr int THREADS = 1024, WAVES = THREADS / 64, RADIX = 256, MAX_PASSES = 16;
template <class Key> struct Tile { static constexpr int ITEMS = sizeof(Key) == 8 ? 8 : 4, SIZE = THREADS * ITEMS; };
static constexpr unsigned long long ST_AGG = 1ull << 62, ST_INCL = 2ull << 62, ST_MASK = 3ull << 62;

template <class Key> __device__ __forceinline__ uint32_t digit_of(Key k, int shift) { return (uint32_t)(k >> shift) & 255u; }

// counts of every digit of every pass, one read of the keys
template <class Key>
__global__ __launch_bounds__(1024) void hist_kernel(const Key *__restrict__ keys, uint32_t n, int firstBit, int nPasses, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t sh[MAX_PASSES][RADIX];
    for (int i = threadIdx.x; i < nPasses * RADIX; i += blockDim.x) (&sh[0][0])[i] = 0u;
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Key k = keys[i];
        for (int p = 0; p < nPasses; ++p) atomicAdd(&sh[p][digit_of<Key>(k, firstBit + 8 * p)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nPasses * RADIX; i += blockDim.x) { const uint32_t c = (&sh[0][0])[i]; if (c) atomicAdd(&hist[i], c); }
}

// hist[p][d] -> first place of digit d in pass p (exclusive running sum per pass)
__global__ __launch_bounds__(256) void starts_kernel(uint32_t *__restrict__ hist, int nPasses)
{
    __shared__ uint32_t sh[RADIX];
    for (int p = 0; p < nPasses; ++p) {
        const uint32_t c = hist[p * RADIX + threadIdx.x];
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int off = 1; off < RADIX; off <<= 1) {
            const uint32_t o = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += o;
            __syncthreads();
        }
        hist[p * RADIX + threadIdx.x] = sh[threadIdx.x] - c;
        __syncthreads();
    }
}

// one pass: pairs (kin, vin) -> (kout, vout), stably by the 8 key bits at `shift`
// MODE (taps of tools/pass_probe.hip and of the tests; the product runs MODE 0): bit 3 = the look-back right after the tile's
// counts are out, before its keys are ordered in LDS (rounds 2-3: 41 ms for five passes over 1.15e9 pairs against 38); bit 1 =
// no look-back (timing only, pairs land in wrong places: 27 ms).
static constexpr int MODE_NOLOOK = 2, MODE_FIRST = 8;
template <class Key, bool PAIRS, int MODE>
__global__ __launch_bounds__(THREADS, 8) void pass_kernel(const Key *__restrict__ kin, const uint32_t *__restrict__ vin, Key *__restrict__ kout,
                                                       uint32_t *__restrict__ vout, uint32_t n, int shift, const uint32_t *__restrict__ start,
                                                       unsigned long long *__restrict__ status, uint32_t *__restrict__ tileCounter, uint32_t nTiles)
{
    constexpr int ITEMS = Tile<Key>::ITEMS, TILE = Tile<Key>::SIZE;
    constexpr bool LATE = (MODE & MODE_FIRST) == 0;
    // one LDS area, three lives: the waves' digit counters while ranking, then the tile's keys in output order, then its payloads
    __shared__ __attribute__((aligned(16))) unsigned char sArea[TILE * sizeof(Key) > WAVES * RADIX * 4 ? TILE * sizeof(Key) : WAVES * RADIX * 4];
    __shared__ uint32_t sBase[RADIX];      // first place of a digit inside the tile
    __shared__ uint32_t sOff[RADIX];       // global place of a digit's first pair of this tile, minus sBase
    __shared__ uint32_t sScan[RADIX / 64];
    __shared__ uint32_t sTile;
    uint32_t (*sCnt)[RADIX] = reinterpret_cast<uint32_t (*)[RADIX]>(sArea);
    Key *sKey = reinterpret_cast<Key *>(sArea);
    uint32_t *sVal = reinterpret_cast<uint32_t *>(sArea);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sTile = atomicAdd(tileCounter, 1u);                  // tiles are numbered in the order the workgroups start
#pragma unroll
    for (int i = 0; i < RADIX / 64; ++i) sCnt[wv][lane + 64 * i] = 0u;
    __syncthreads();
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)sTile);
    // wave-striped: row i of a wave = 64 consecutive pairs.  The tile's first pair is a scalar address, a thread adds one 32-bit
    // offset and the row's constant: eight loads from one address register
    const Key *__restrict__ kt = kin + (uint64_t)tile * TILE;
    const uint32_t *__restrict__ vt = vin + (uint64_t)tile * TILE;
    const uint32_t base = (uint32_t)wv * (64 * ITEMS) + (uint32_t)lane;
    const uint32_t left = (uint64_t)tile * TILE < (uint64_t)n ? (uint32_t)((uint64_t)n - (uint64_t)tile * TILE) : 0u;
    const uint32_t count = left < (uint32_t)TILE ? left : (uint32_t)TILE;   // pairs of this tile
    Key k[ITEMS];
    uint32_t rank[ITEMS];                                             // rank: inside the wave first, then the pair's place in the tile
    uint32_t okMask = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const bool ok = base + (uint32_t)i * 64u < count;
        okMask |= ok ? (1u << i) : 0u;
        k[i] = ok ? kt[base + (uint32_t)i * 64u] : (Key)0;
    }
    // ---- rank inside the wave: pairs of one row with the same digit find each other by ballots; the first of them books the
    // digit's running count of this wave (LDS, wave-private: a wave's LDS instructions execute in order)
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t d = digit_of<Key>(k[i], shift);
        const bool ok = (okMask >> i) & 1u;
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long mb = __ballot(bit);
            m &= bit ? mb : ~mb;
        }
        uint32_t old = 0;
        const int leader = __ffsll((long long)m) - 1;                // (ok: the lane itself is in m)
        if (ok && lane == leader) { old = sCnt[wv][d]; sCnt[wv][d] = old + (uint32_t)__popcll(m); }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        old = __shfl(old, leader < 0 ? 0 : leader);
        rank[i] = old + (uint32_t)__popcll(m & below);
        __builtin_amdgcn_sched_barrier(0);                            // (rows one after the other: interleaved, their temporaries cost registers the payloads need)
    }
    uint32_t v[ITEMS];
    __syncthreads();
    // ---- per digit: the waves' counts become running sums, the tile's total is published, the digits' places in the tile
    // and -- by looking back over the earlier tiles -- in the output follow
    uint32_t total = 0, digitBase = 0;
    if (tid < RADIX) {
        for (int w = 0; w < WAVES; ++w) { const uint32_t c = sCnt[w][tid]; sCnt[w][tid] = total; total += c; }
        __hip_atomic_store(&status[(size_t)tile * RADIX + tid], (tile == 0 ? ST_INCL : ST_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t incl = total;                                        // running sum over the 256 digits: inside the wave ...
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) sScan[wv] = incl;
        digitBase = incl - total;
    }
    __syncthreads();
    if (tid < RADIX) {
        for (int w = 0; w < wv; ++w) digitBase += sScan[w];           // ... and over the four waves that hold the digits
        sBase[tid] = digitBase;
    }
    // The tile's own counts are out (ST_AGG); where its digits go in the output -- the pairs with the digit in all earlier
    // tiles, found by looking back over them -- is only needed when the keys leave.  So the tile first orders its keys in LDS
    // and asks for its payloads, and looks back then: the tiles before it have had that time to publish, and the payloads are
    // on their way while it waits.  (Rounds 2-3 looked back first.)  What the walk is like, from tools/pass_probe.hip: the
    // tiles before are mostly as far as this one -- counts out, sums not -- so it is 15 tiles long on average (69 at most) and
    // hardly ever polls an empty word; it still takes a quarter of the pass (38 ms for five passes over 1.15e9 pairs, 27 with
    // the walk cut out).  Neither more loads in flight (4 / 8 / 16 tiles asked for at once: 38.0 / 39.1 / 84 ms; four lanes
    // per digit: 52 ms) nor fewer bytes (16-bit counts, four digits to a word, read by one wavefront, the 64-bit sums fetched
    // where the walk ends: 61 ms -- the extra round trip at the end delays every tile behind) did better than this.
    auto look_back = [&]() {
        if (tid < RADIX) {
            unsigned long long prev = 0;
            if constexpr ((MODE & MODE_NOLOOK) == 0)
                for (int64_t t = (int64_t)tile - 1; t >= 0; --t) {
                    unsigned long long sv;
                    do { sv = __hip_atomic_load(&status[(size_t)t * RADIX + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((sv & ST_MASK) == 0ull);
                    prev += sv & ~ST_MASK;
                    if ((sv & ST_MASK) == ST_INCL) break;
                }
            if (tile != 0) __hip_atomic_store(&status[(size_t)tile * RADIX + tid], ST_INCL | (prev + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sOff[tid] = start[tid] + (uint32_t)prev - digitBase;
        }
    };
    if constexpr (!LATE) look_back();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { const uint32_t d = digit_of<Key>(k[i], shift); rank[i] += sBase[d] + sCnt[wv][d]; }
    __syncthreads();                                                  // the counters are dead: their area takes the keys
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) if ((okMask >> i) & 1u) sKey[rank[i]] = k[i];
    if constexpr (PAIRS && LATE) {                                    // (the keys' registers are free now)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) v[i] = ((okMask >> i) & 1u) ? vt[base + (uint32_t)i * 64u] : 0u;
    }
    if constexpr (LATE) look_back();
    __syncthreads();
    unsigned long long digits = 0;                                    // digit of the pair this thread writes in round j, 8 bits each
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {                                 // neighbouring threads write neighbouring pairs of a run
        const uint32_t p = (uint32_t)j * THREADS + tid;
        if (p < count) {
            const Key kk = sKey[p];
            const uint32_t d = digit_of<Key>(kk, shift);
            digits |= (unsigned long long)d << (8 * j);
            kout[sOff[d] + p] = kk;
        }
    }
    if constexpr (!PAIRS) return;
    // the payloads follow the same way (loaded only now: the keys' registers are free, and the other workgroup of the CU
    // covers the wait)
    if constexpr (!LATE) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) v[i] = ((okMask >> i) & 1u) ? vt[base + (uint32_t)i * 64u] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) if ((okMask >> i) & 1u) sVal[rank[i]] = v[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t p = (uint32_t)j * THREADS + tid;
        if (p < count) vout[sOff[(uint32_t)(digits >> (8 * j)) & 255u] + p] = sVal[p];
    }
}

template <class Key> inline uint32_t tiles_of(uint32_t n) { return (uint32_t)(((uint64_t)n + Tile<Key>::SIZE - 1) / Tile<Key>::SIZE); }
// scratch: digit starts of every pass, the tile counter, the look-back words of one pass
template <class Key> inline size_t scratch_bytes(uint64_t n)
{
    return (size_t)MAX_PASSES * RADIX * 4 + 256 + (size_t)tiles_of<Key>((uint32_t)n) * RADIX * 8;
}

// Stable sort of n pairs by the key bits [firstBit, firstBit + nBits), nBits a multiple of 8.  (kA, vA) holds the input and is
// overwritten; the result lies in (kB, vB) after an odd number of passes, else back in (kA, vA): *kRes / *vRes say where.
// vA = vB = NULL: keys only.
template <class Key>
inline hipError_t sort_pairs(Key *kA, uint32_t *vA, Key *kB, uint32_t *vB, uint32_t n, int firstBit, int nBits, void *scratch, hipStream_t stream,
                             Key **kRes, uint32_t **vRes, int variant = 0)
{
    const int nPasses = nBits / 8;
    *kRes = kA; if (vRes) *vRes = vA;
    if (n == 0 || nPasses == 0) return hipSuccess;
    if (nPasses > MAX_PASSES || nBits % 8) return hipErrorInvalidValue;
    uint32_t *hist = static_cast<uint32_t *>(scratch);
    uint32_t *counter = hist + MAX_PASSES * RADIX;
    unsigned long long *status = reinterpret_cast<unsigned long long *>(static_cast<char *>(scratch) + (size_t)MAX_PASSES * RADIX * 4 + 256);
    const uint32_t nTiles = tiles_of<Key>(n);
    hipError_t e;
    if ((e = hipMemsetAsync(hist, 0, (size_t)MAX_PASSES * RADIX * 4 + 256, stream)) != hipSuccess) return e;
    const unsigned hb = (unsigned)((nTiles < 2048u) ? (nTiles ? nTiles : 1u) : 2048u);
    hist_kernel<Key><<<hb, 1024, 0, stream>>>(kA, n, firstBit, nPasses, hist);
    starts_kernel<<<1, 256, 0, stream>>>(hist, nPasses);
    Key *ki = kA, *ko = kB; uint32_t *vi = vA, *vo = vB;
    for (int p = 0; p < nPasses; ++p) {
        if ((e = hipMemsetAsync(status, 0, (size_t)nTiles * RADIX * 8, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(counter, 0, 32, stream)) != hipSuccess) return e;
#define KASA_PASS(PAIRS_, MODE_) pass_kernel<Key, PAIRS_, MODE_><<<nTiles, THREADS, 0, stream>>>(ki, vi, ko, vo, n, firstBit + 8 * p, hist + p * RADIX, status, counter, nTiles)
        if (!vA) KASA_PASS(false, 0);
        else if (variant == MODE_FIRST) KASA_PASS(true, MODE_FIRST);
        else if (variant == MODE_NOLOOK) KASA_PASS(true, MODE_NOLOOK);
        else KASA_PASS(true, 0);
#undef KASA_PASS
        Key *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
    *kRes = ki; if (vRes) *vRes = vi;
    return hipGetLastError();
}

} // namespace kasa_radix

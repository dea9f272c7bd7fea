// kasa_replay.h -- the score stage for VERY LONG reads (a contig, a chromosome: millions of k-mers under one read id).
// Included by kasa_hip.hip behind score_kernel; uses its ScoreArgs, QueryRec, kasa_ctx.
//
// The reference streams any sequence through compareWithDatabase at merge speed (source/modes/Compare.hpp:747-1043; a long
// sequence arrives in pieces, source/modes/Read.hpp:437-443,678-695, and its scores are carried from batch to batch).  The
// general kernel above replays a read's events in flush order with ONE wavefront -- 0.3-0.6 M k-mers/s, the chain of
// dependent loads of one query after the other: a 9.6 Mbp contig took 37 s of device time (round 5).  But the
// order is known before anything is added: an event is (flush position F, level k) of one query, events are replayed in
// ascending (F, k) -- exactly what the pending window of score_kernel produces, equal (F, k) being the same group hit once
// more -- and the float sums that depend on it are chains PER (read, taxon): M[r][t] += w_k * (1 / |T_k|), one add per hit
// (Compare.hpp:516-532, :924).  So:
//   1. esr_count_kernel / esr_emit_kernel: every query of the listed reads, by ALL wavefronts of the chip, turns its record
//      into events {read, taxon, F_k, k} -> float32 addend (the flush positions made on the way, as flush_positions_kernel
//      makes them);
//   2. one radix sort of the events by (read, taxon, F, k) (the library's: a cold path);
//   3. esr_chain_*_kernel: a chain = the events of one (read, taxon), added front to back in float32 -- short chains a lane
//      each, long ones a wavefront each (the addends staged in LDS a kilobyte ahead of the adds);
//   4. esr_rows_kernel: the chains of a read in ascending taxon order ARE its row.
// What bounds it is the longest chain (a taxon that every k-mer of a 20 M-k-mer read touches: 1e8 dependent float adds of one
// lane, a few hundred ms); everything else is parallel work of the kind the batch does anyway.  Both record widths: the
// events of 64-byte records (a 128-bit index: the profile is not the group stage's) are also added to the profile tables,
// one exact integer add per event (esr_addend_kernel).
#pragma once

static constexpr uint32_t ESR_MIN_KMERS = 16384;      // reads of the general kernel's list with at least this many k-mers are replayed from sorted events
static constexpr uint32_t ESR_BIG_CHAIN = 256;        // events from which a chain is a wavefront's
static constexpr int ESR_CHUNK = 1024;                // addends a wavefront stages per step
static constexpr uint64_t ESR_ROUND_EVENTS = 1ull << 31;   // events sorted at once (32-bit chain starts; 24 bytes of buffers each)

// kmerOff-relative geometry of the listed reads: cnt[i] = queries of read list[i] (list NULL: read i)
__global__ void esr_iota_kernel(uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// reads of `list` with at least minK k-mers -> longList, the others -> shortList (order: as the atomics fall; no result depends on it)
__global__ void esr_split_kernel(const uint32_t *__restrict__ list, uint32_t nList, const uint64_t *__restrict__ kmerOff, uint32_t minK,
                                 uint32_t *__restrict__ longList, uint32_t *__restrict__ shortList, uint32_t *__restrict__ counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nList) return;
    const uint32_t r = list[i];
    const uint64_t n = kmerOff[r + 1] - kmerOff[r];
    if (n >= (uint64_t)minK) longList[atomicAdd(&counts[0], 1u)] = r; else shortList[atomicAdd(&counts[1], 1u)] = r;
}

// flat query index g of the listed reads -> (list entry, query of the read); qOff = running sum of the listed reads' queries
__device__ __forceinline__ uint32_t esr_owner(const uint64_t *__restrict__ qOff, uint32_t nList, uint64_t g)
{
    uint32_t lo = 0, hi = nList;                       // qOff[lo] <= g < qOff[hi]
    while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (qOff[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

// events a query yields: one per (segment, level of the segment)
template <int RW>
__global__ __launch_bounds__(256) void esr_count_kernel(ScoreArgs A, const uint32_t *__restrict__ list, uint32_t nList, const uint64_t *__restrict__ qOff,
                                                        uint64_t nq, uint32_t *__restrict__ evCnt)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nq; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t wi = esr_owner(qOff, nList, g);
        const uint64_t slot = A.kmerOff[list[wi]] + (g - qOff[wi]);
        const uint32_t *w = A.rec + slot * A.recCW;
        uint32_t n = 0;
        if ((w[2] & 31u) != 0u) {
            const uint32_t nseg = rec_nseg<RW>(w, A.pool);
            for (uint32_t i = 0; i < nseg; ++i) { const uint32_t s = rec_seg<RW>(w, A.pool, nseg, i); n += (s >> 27) - ((s >> 22) & 31u) + 1u; }
        }
        evCnt[g] = n;
    }
}

// events per listed read, from the running sum over the queries
__global__ void esr_read_events_kernel(const uint64_t *__restrict__ qOff, uint32_t nList, const uint64_t *__restrict__ evOff, uint64_t *__restrict__ readEv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nList) readEv[i] = evOff[qOff[i]];
}

// One WAVEFRONT per query: its flush positions (as flush_positions_kernel: the rest of p's tile, then the per-tile table), then
// one event per (segment, level): key = read (its place in the round) | taxon | F | k -- ascending keys = (read, taxon) chains in
// the reference's flush order -- and the addend w_k * (1 / |T_k|) in float32 (Compare.hpp:923-924).
template <class Key, int RW>
__global__ __launch_bounds__(256) void esr_emit_kernel(ScoreArgs A, const uint32_t *__restrict__ list, uint32_t nList, const uint64_t *__restrict__ qOff,
    uint32_t w0, uint64_t q0, uint64_t q1, const uint64_t *__restrict__ evOff, const Key *__restrict__ qKmer, const uint8_t *__restrict__ depth,
    const uint32_t *__restrict__ tileNext, uint32_t nTiles, int taxBits, uint64_t *__restrict__ keys, uint32_t *__restrict__ sizes)
{
    __shared__ uint32_t sF[4][32], sN[4][33];                       // per wavefront: F of every level; |T| of every level (marks, then their running sum)
    const int nK = A.kHigh - A.kLow + 1;
    const uint32_t allLv = (1u << nK) - 1u;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t evBase = evOff[q0];
    for (uint64_t g = q0 + (uint64_t)blockIdx.x * 4u + (uint64_t)wv; g < q1; g += (uint64_t)gridDim.x * 4u) {
        const uint32_t wi = esr_owner(qOff, nList, g);
        const uint64_t slot = A.kmerOff[list[wi]] + (g - qOff[wi]);
        const uint32_t *w = A.rec + slot * A.recCW;
        const uint32_t p = w[0];
        const int d = (int)(w[2] & 31u);
        if (d == 0) continue;                                          // (uniform)
        uint32_t myF = NOPOS;                                          // lane lv holds F of level lv
        {
            const uint32_t tile = p / TILE;
            const uint32_t tileEnd = ((uint64_t)(tile + 1) * TILE < A.nQ) ? (tile + 1) * TILE : A.nQ;
            uint32_t todo = allLv & ~((1u << (A.kHigh - d)) - 1u);     // levels kLow..d
            for (uint32_t b0 = p + 1; b0 < tileEnd && todo; b0 += 64) {
                const uint32_t pp = b0 + lane;
                uint32_t m = 0;
                if (pp < tileEnd) {
                    const int ql = lcp_letters<Key>(qKmer[pp - 1], qKmer[pp]);
                    m = special_mask(ql, (int)depth[pp], A.kHigh, allLv);
                }
                for (int lv = 0; lv < nK; ++lv) {
                    if (!((todo >> lv) & 1u)) continue;
                    const unsigned long long b = __ballot((m >> lv) & 1u);
                    if (b) { if (lane == lv) myF = b0 + (uint32_t)(__ffsll((long long)b) - 1); todo &= ~(1u << lv); }
                }
            }
            if (lane < nK && ((todo >> lane) & 1u)) myF = tileNext[(size_t)lane * nTiles + tile];
        }
        if (lane < 32) { sF[wv][lane] = myF; sN[wv][lane] = 0u; }
        if (lane == 32) sN[wv][32] = 0u;
        LDS_WAVE_SYNC();
        // |T_k| of every level of the query: +1 / -1 at the ends of each segment's level range, then a running sum (the records
        // of a narrow index carry them, those of a 128-bit index do not: one way for both)
        const uint32_t nseg = rec_nseg<RW>(w, A.pool);
        for (uint32_t b0 = 0; b0 < nseg; b0 += 64) {
            const uint32_t i = b0 + lane;
            if (i < nseg) {
                const uint32_t sg = rec_seg<RW>(w, A.pool, nseg, i);
                atomicAdd(&sN[wv][A.kHigh - (int)(sg >> 27)], 1u);
                atomicSub(&sN[wv][A.kHigh - (int)((sg >> 22) & 31u) + 1], 1u);
            }
        }
        LDS_WAVE_SYNC();
        {
            uint32_t v = lane < 32 ? sN[wv][lane] : 0u;
            v = wave_incl_sum(v);
            LDS_WAVE_SYNC();
            if (lane < 32) sN[wv][lane] = v;
        }
        LDS_WAVE_SYNC();
        const uint64_t readKey = (uint64_t)(wi - w0) << (37 + taxBits);
        uint64_t at = evOff[g] - evBase;
        for (uint32_t b0 = 0; b0 < nseg; b0 += 64) {
            const uint32_t i = b0 + lane;
            const uint32_t sg = i < nseg ? rec_seg<RW>(w, A.pool, nseg, i) : 0u;
            const uint32_t kFirst = (sg >> 22) & 31u, kLast = sg >> 27;
            const uint32_t cnt = i < nseg ? kLast - kFirst + 1u : 0u;
            const uint32_t incl = wave_incl_sum(cnt);
            uint64_t o = at + incl - cnt;
            const uint64_t tk = readKey | ((uint64_t)(sg & SEG_TAX_MASK) << 37);
            for (uint32_t k = kFirst; i < nseg && k <= kLast; ++k, ++o) {
                const int lv = A.kHigh - (int)k;
                keys[o] = tk | ((uint64_t)sF[wv][lv] << 5) | (uint64_t)k;
                sizes[o] = sN[wv][lv];
            }
            at += lane_value<63>(incl);
        }
        LDS_WAVE_SYNC();
    }
}

// after the sort: the payload |T_k| becomes the event's float32 addend w_k * (1 / |T_k|) (Compare.hpp:923-924), in place; records of
// a 128-bit index (the profile is not the group stage's there) also add the event to the profile tables: countAll[k][t] += 1 / |T|,
// countUnique[k][t] += 1 iff |T| = 1 (Compare.hpp:922-925; exact integer limbs, the order does not matter)
__global__ void esr_addend_kernel(const uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t n, int taxBits, ScoreArgs A, int addProfile)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const int k = (int)(key & 31u);
    const uint32_t sz = vals[i];
    vals[i] = __float_as_uint(event_score(k, sz));
    if (addProfile) profile_add(A, A.kHigh - k, (uint32_t)(key >> 37) & ((1u << taxBits) - 1u), sz, 1ull);
}

// chain heads of the sorted events: a new (read, taxon)
__global__ void esr_heads_kernel(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ head)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0u || (keys[i] >> 37) != (keys[i - 1] >> 37)) ? 1u : 0u;
}
__global__ void esr_starts_kernel(const uint64_t *__restrict__ keys, uint32_t n, const uint32_t *__restrict__ headRank, uint32_t *__restrict__ chainStart, uint32_t nChains)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (i == 0u || (keys[i] >> 37) != (keys[i - 1] >> 37))) chainStart[headRank[i]] = i;
    if (i == 0u) chainStart[nChains] = n;
}

// the float chain of one (read, taxon): Compare.hpp:528-530, one add per hit, in flush order
__global__ void esr_chain_small_kernel(const float *__restrict__ vals, const uint32_t *__restrict__ chainStart, uint32_t nChains, float *__restrict__ chainScore)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nChains) return;
    const uint32_t s = chainStart[c], e = chainStart[c + 1];
    if (e - s >= ESR_BIG_CHAIN) return;
    float x = 0.0f;
    for (uint32_t i = s; i < e; ++i) x = __fadd_rn(x, vals[i]);
    chainScore[c] = x;
}
// ... a long one: the wavefront loads ESR_CHUNK addends at a time (coalesced: lane l holds addends u * 64 + l of the chunk, the
// next chunk on its way while this one is added).  The chain itself is ONE instruction per addend: x = rotate_right_1(x) + a,
// a DPP add over the whole wavefront -- the true partial sum travels from lane to lane (at step t it sits in lane t mod 64,
// which adds ITS addend; what the other lanes compute is never looked at) and after every 64 steps it is back in lane 63.
// (Out of LDS, four addends per read: 11.5 ns per addend -- the 53 M-event chain of a contig that is one genome 32 times over
// took 611 ms; v_readlane + v_add: 6.8 ns.)  Addends beyond the chain's end are +0.0f: they leave the (positive) sum as it is.
__global__ __launch_bounds__(64) void esr_chain_big_kernel(const float *__restrict__ vals, const uint32_t *__restrict__ chainStart, uint32_t nChains,
                                                           const uint32_t *__restrict__ bigList, uint32_t nBig, float *__restrict__ chainScore)
{
    const int lane = threadIdx.x;
    constexpr int PER = ESR_CHUNK / 64;
    for (uint32_t bi = blockIdx.x; bi < nBig; bi += gridDim.x) {
        const uint32_t c = bigList[bi];
        const uint32_t s = chainStart[c], e = chainStart[c + 1];
        float x = 0.0f;
        float reg[PER], cur[PER];
        auto load = [&](uint32_t b0) {
#pragma unroll
            for (int u = 0; u < PER; ++u) { const uint32_t i = b0 + (uint32_t)(u * 64 + lane); reg[u] = i < e ? vals[i] : 0.0f; }
        };
        load(s);
        for (uint32_t b0 = s; b0 < e; b0 += ESR_CHUNK) {
#pragma unroll
            for (int u = 0; u < PER; ++u) cur[u] = reg[u];
            if (b0 + ESR_CHUNK < e) load(b0 + ESR_CHUNK);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
#pragma unroll
                for (int l = 0; l < 64; ++l) {
                    const float prev = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x13C /* wave_ror:1 */, 0xf, 0xf, false));
                    x = __fadd_rn(prev, cur[u]);
                }
            }
        }
        const float total = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), 63));
        if (lane == 0) chainScore[c] = total;
    }
}
__global__ void esr_big_list_kernel(const uint32_t *__restrict__ chainStart, uint32_t nChains, uint32_t *__restrict__ bigList, uint32_t *__restrict__ nBig)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nChains && chainStart[c + 1] - chainStart[c] >= ESR_BIG_CHAIN) bigList[atomicAdd(nBig, 1u)] = c;
}

// the rows: chains are in (read, taxon) order -- a read's chains, as they lie, are its row {taxon, score}
__global__ void esr_rows_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ chainStart, uint32_t nChains, const float *__restrict__ chainScore,
                                const uint32_t *__restrict__ list, uint32_t w0, int taxBits, ScoreArgs A, unsigned long long *__restrict__ rowBase)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nChains) return;
    const unsigned long long base = *rowBase;
    if (base + nChains > (unsigned long long)A.stCap) return;          // (the host sees the cursor's demand and runs the stage again)
    const uint64_t k = keys[chainStart[c]];
    const uint32_t tx = (uint32_t)(k >> 37) & ((1u << taxBits) - 1u), rl = (uint32_t)(k >> (37 + taxBits));
    A.st[base + c] = make_uint2(tx, __float_as_uint(chainScore[c]));
    const uint32_t r = list[w0 + rl];
    const bool first = c == 0u || (uint32_t)(keys[chainStart[c - 1]] >> (37 + taxBits)) != rl;
    if (first) A.rowPos[r] = (uint32_t)(base + c);
    atomicAdd(&A.rowLen[r], 1u);
}
__global__ void esr_row_base_kernel(unsigned long long *__restrict__ stCursor, uint32_t nChains, unsigned long long *__restrict__ rowBase,
                                    const uint32_t *__restrict__ list, uint32_t w0, uint32_t w1, uint32_t *__restrict__ rowPos, uint32_t *__restrict__ rowLen)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) *rowBase = atomicAdd(stCursor, (unsigned long long)nChains);
    if (w0 + i < w1) { rowPos[list[w0 + i]] = 0u; rowLen[list[w0 + i]] = 0u; }   // (a read without a match keeps an empty row)
}

// The reads of A.list (nSlow of them; NULL: all reads) with at least minK k-mers are scored here; what is left for the general
// kernel comes back in *listOut / *nOut.  counters: the context's misc words (this stage uses [72..79]).
static int esr_stage(kasa_ctx *c, ScoreArgs &A, uint32_t nSlow, uint32_t minK, const uint32_t **listOut, uint32_t *nOut, uint32_t *counters)
{
    int rc;
    const uint32_t nReads = (uint32_t)c->nReads;
    const int nK = c->nK;
    *listOut = A.list; *nOut = nSlow;
    c->lastReplayReads = 0; c->lastReplayEvents = 0;
    if (nSlow == 0) return KASA_OK;
    if (c->maxCnt < minK) return KASA_OK;                                  // no read of the batch is that long
    if ((rc = c->esrLong.reserve((size_t)nSlow * 4 + 64)) || (rc = c->esrShort.reserve((size_t)nSlow * 4 + 64))) return rc;
    const uint32_t *list = A.list;
    if (!list) {                                                           // (the general kernel over all reads: forceSlowScore)
        if ((rc = c->esrIota.reserve((size_t)nReads * 4 + 64))) return rc;
        esr_iota_kernel<<<blocks_for(nReads, 256), 256, 0, c->stream>>>(c->esrIota.as<uint32_t>(), nReads);
        list = c->esrIota.as<uint32_t>();
    }
    uint32_t *cnt2 = counters + 72;
    HIPCHK(hipMemsetAsync(cnt2, 0, 8, c->stream));
    esr_split_kernel<<<blocks_for(nSlow, 256), 256, 0, c->stream>>>(list, nSlow, A.kmerOff, minK, c->esrLong.as<uint32_t>(), c->esrShort.as<uint32_t>(), cnt2);
    uint32_t h2[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h2, cnt2, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t nLong = h2[0], nShort = h2[1];
    if (nLong == 0) return KASA_OK;
    const uint32_t *longList = c->esrLong.as<uint32_t>();
    // queries of the long reads (running sum), events of every query (running sum)
    if ((rc = c->esrQOff.reserve(((size_t)nLong + 1) * 8 + 64)) || (rc = c->esrReadEv.reserve(((size_t)nLong + 1) * 8 + 64))) return rc;
    list_counts_kernel<<<blocks_for((uint64_t)nLong + 1, 256), 256, 0, c->stream>>>(longList, nLong, A.kmerOff, c->esrQOff.as<uint64_t>());
    size_t tmpBytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, c->esrQOff.as<uint64_t>(), c->esrQOff.as<uint64_t>(), (uint64_t)0, (size_t)nLong + 1, rocprim::plus<uint64_t>(), c->stream));
    if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
    HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, c->esrQOff.as<uint64_t>(), c->esrQOff.as<uint64_t>(), (uint64_t)0, (size_t)nLong + 1, rocprim::plus<uint64_t>(), c->stream));
    uint64_t nq = 0;
    HIPCHK(hipMemcpyAsync(&nq, c->esrQOff.as<uint64_t>() + nLong, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = c->esrEvCnt.reserve((nq + 1) * 4 + 64)) || (rc = c->esrEvOff.reserve((nq + 1) * 8 + 64))) return rc;
    HIPCHK(hipMemsetAsync(c->esrEvCnt.as<uint32_t>() + nq, 0, 4, c->stream));
    if (c->recWords() == 8) esr_count_kernel<8><<<std::min<unsigned>(blocks_for(nq, 256), 256u * 64u), 256, 0, c->stream>>>(A, longList, nLong, c->esrQOff.as<uint64_t>(), nq, c->esrEvCnt.as<uint32_t>());
    else esr_count_kernel<16><<<std::min<unsigned>(blocks_for(nq, 256), 256u * 64u), 256, 0, c->stream>>>(A, longList, nLong, c->esrQOff.as<uint64_t>(), nq, c->esrEvCnt.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, c->esrEvCnt.as<uint32_t>(), c->esrEvOff.as<uint64_t>(), (uint64_t)0, (size_t)nq + 1, rocprim::plus<uint64_t>(), c->stream));
    if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
    HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, c->esrEvCnt.as<uint32_t>(), c->esrEvOff.as<uint64_t>(), (uint64_t)0, (size_t)nq + 1, rocprim::plus<uint64_t>(), c->stream));
    esr_read_events_kernel<<<blocks_for((uint64_t)nLong + 1, 256), 256, 0, c->stream>>>(c->esrQOff.as<uint64_t>(), nLong, c->esrEvOff.as<uint64_t>(), c->esrReadEv.as<uint64_t>());
    std::vector<uint64_t> readEv((size_t)nLong + 1), qOffH((size_t)nLong + 1);
    HIPCHK(hipMemcpyAsync(readEv.data(), c->esrReadEv.p, ((size_t)nLong + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(qOffH.data(), c->esrQOff.p, ((size_t)nLong + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // rounds of consecutive long reads: as many events as the buffers hold (24 bytes each + the sort's own), as many reads as
    // the key has bits for
    int taxBits = 1;
    while ((1ull << taxBits) < (uint64_t)A.nTaxa) ++taxBits;
    const int readBits = std::min(20, 64 - 37 - taxBits);
    size_t freeB = 0, totalB = 0;
    HIPCHK(hipMemGetInfo(&freeB, &totalB));
    const uint64_t have = c->esrKeyA.cap + c->esrKeyB.cap + c->esrValA.cap + c->esrValB.cap;
    uint64_t budget = std::min<uint64_t>(ESR_ROUND_EVENTS, ((uint64_t)freeB * 3 / 4 + have) / 40);
    const uint64_t capEnv = getenv("KASA_ESR_ROUND_EVENTS") ? strtoull(getenv("KASA_ESR_ROUND_EVENTS"), nullptr, 10) : 0;   // (tests: several rounds on small inputs)
    if (capEnv) budget = std::min<uint64_t>(budget, capEnv);
    const uint32_t nTiles = (uint32_t)((c->nQ + TILE - 1) / TILE);
    unsigned long long *rowBase = reinterpret_cast<unsigned long long *>(counters + 74);
    uint32_t *nBigDev = counters + 76;
    std::vector<uint32_t> back;                                            // long reads that do not fit a round: the general kernel's after all
    uint32_t w0 = 0;
    while (w0 < nLong) {
        uint32_t w1 = w0;
        while (w1 < nLong && w1 - w0 < (1u << readBits) && readEv[w1 + 1] - readEv[w0] <= budget) ++w1;
        if (w1 == w0) { back.push_back(w0); ++w0; continue; }            // a single read beyond the budget
        const uint64_t E = readEv[w1] - readEv[w0];
        const uint64_t q0 = qOffH[w0], q1 = qOffH[w1];
        if (E > 0) {
            if ((rc = c->esrKeyA.reserve(E * 8 + 256)) || (rc = c->esrKeyB.reserve(E * 8 + 256)) || (rc = c->esrValA.reserve(E * 4 + 64)) || (rc = c->esrValB.reserve(E * 4 + 64))) return rc;
            const unsigned eblocks = (unsigned)std::min<uint64_t>((q1 - q0 + 3) / 4, 256u * 32u);
#define KASA_ESR_EMIT(KEY, RWV) esr_emit_kernel<KEY, RWV><<<eblocks, 256, 0, c->stream>>>(A, longList, nLong, c->esrQOff.as<uint64_t>(), w0, q0, q1, c->esrEvOff.as<uint64_t>(), \
                c->keys<KEY>(), c->depth.as<uint8_t>(), c->tileNext.as<uint32_t>(), nTiles, taxBits, c->esrKeyA.as<uint64_t>(), c->esrValA.as<uint32_t>())
            if (c->ix->wide) { if (c->recWords() == 8) KASA_ESR_EMIT(key128, 8); else KASA_ESR_EMIT(key128, 16); }
            else { if (c->recWords() == 8) KASA_ESR_EMIT(uint64_t, 8); else KASA_ESR_EMIT(uint64_t, 16); }   // (a 64-bit index with more than eight levels has 64-byte records too)
#undef KASA_ESR_EMIT
            HIPCHK(hipGetLastError());
            int rb = 0;
            while ((1u << rb) < w1 - w0) ++rb;
            const unsigned endBit = (unsigned)(37 + taxBits + rb);
            HIPCHK(rocprim::radix_sort_pairs(nullptr, tmpBytes, c->esrKeyA.as<uint64_t>(), c->esrKeyB.as<uint64_t>(), c->esrValA.as<uint32_t>(), c->esrValB.as<uint32_t>(), (size_t)E, 0u, endBit, c->stream));   // (the addends travel as 32-bit words: the query sort's fallback is the same instantiation of the library's sort)
            if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
            HIPCHK(rocprim::radix_sort_pairs(c->sortTmp.p, tmpBytes, c->esrKeyA.as<uint64_t>(), c->esrKeyB.as<uint64_t>(), c->esrValA.as<uint32_t>(), c->esrValB.as<uint32_t>(), (size_t)E, 0u, endBit, c->stream));   // (the addends travel as 32-bit words: the query sort's fallback is the same instantiation of the library's sort)
            const uint32_t n = (uint32_t)E;
            esr_addend_kernel<<<blocks_for(n, 256), 256, 0, c->stream>>>(c->esrKeyB.as<uint64_t>(), c->esrValB.as<uint32_t>(), n, taxBits, A, A.addProfile);
            HIPCHK(hipGetLastError());
            if (!A.wantPerRead) { c->lastReplayEvents += E; c->lastReplayReads += w1 - w0; w0 = w1; continue; }   // (a profile-only run of a 128-bit index: no rows)
            // chains: heads -> ranks (a running sum, in the dead key buffer) -> starts
            uint32_t *head = c->esrKeyA.as<uint32_t>(), *headRank = head + ((size_t)n + 16);
            esr_heads_kernel<<<blocks_for(n, 256), 256, 0, c->stream>>>(c->esrKeyB.as<uint64_t>(), n, head);
            HIPCHK(rocprim::exclusive_scan(nullptr, tmpBytes, head, headRank, 0u, (size_t)n, rocprim::plus<uint32_t>(), c->stream));
            if ((rc = c->sortTmp.reserve(tmpBytes))) return rc;
            HIPCHK(rocprim::exclusive_scan(c->sortTmp.p, tmpBytes, head, headRank, 0u, (size_t)n, rocprim::plus<uint32_t>(), c->stream));
            uint32_t lastRank = 0, lastHead = 0;
            HIPCHK(hipMemcpyAsync(&lastRank, headRank + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(&lastHead, head + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            const uint32_t nChains = lastRank + lastHead;
            if ((rc = c->esrChain.reserve(((size_t)nChains + 1) * 4 + 64)) || (rc = c->esrChainScore.reserve((size_t)nChains * 4 + 64)) || (rc = c->esrBig.reserve((size_t)nChains * 4 + 64))) return rc;
            esr_starts_kernel<<<blocks_for(n, 256), 256, 0, c->stream>>>(c->esrKeyB.as<uint64_t>(), n, headRank, c->esrChain.as<uint32_t>(), nChains);
            HIPCHK(hipMemsetAsync(nBigDev, 0, 4, c->stream));
            esr_big_list_kernel<<<blocks_for(nChains, 256), 256, 0, c->stream>>>(c->esrChain.as<uint32_t>(), nChains, c->esrBig.as<uint32_t>(), nBigDev);
            uint32_t nBig = 0;
            HIPCHK(hipMemcpyAsync(&nBig, nBigDev, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            esr_chain_small_kernel<<<blocks_for(nChains, 256), 256, 0, c->stream>>>(c->esrValB.as<float>(), c->esrChain.as<uint32_t>(), nChains, c->esrChainScore.as<float>());
            if (nBig) esr_chain_big_kernel<<<std::min<uint32_t>(nBig, 256u * 16u), 64, 0, c->stream>>>(c->esrValB.as<float>(), c->esrChain.as<uint32_t>(), nChains, c->esrBig.as<uint32_t>(), nBig, c->esrChainScore.as<float>());
            esr_row_base_kernel<<<blocks_for(w1 - w0, 256), 256, 0, c->stream>>>(A.stCursor, nChains, rowBase, longList, w0, w1, A.rowPos, A.rowLen);
            esr_rows_kernel<<<blocks_for(nChains, 256), 256, 0, c->stream>>>(c->esrKeyB.as<uint64_t>(), c->esrChain.as<uint32_t>(), nChains, c->esrChainScore.as<float>(), longList, w0, taxBits, A, rowBase);
            HIPCHK(hipGetLastError());
            c->lastReplayEvents += E;
        }
        c->lastReplayReads += w1 - w0;
        w0 = w1;
    }
    (void)nK;
    if (!back.empty()) {                                                   // (their ids go behind the short list)
        std::vector<uint32_t> longH(nLong);
        HIPCHK(hipMemcpy(longH.data(), c->esrLong.p, (size_t)nLong * 4, hipMemcpyDeviceToHost));
        for (uint32_t wi : back) { HIPCHK(hipMemcpy(c->esrShort.as<uint32_t>() + nShort, &longH[wi], 4, hipMemcpyHostToDevice)); ++nShort; }
    }
    // the event buffers of a contig are tens of gigabytes: a context that goes on with ordinary batches gets them back
    if (c->esrKeyA.cap + c->esrKeyB.cap + c->esrValA.cap + c->esrValB.cap > ((size_t)2 << 30)) {
        HIPCHK(hipStreamSynchronize(c->stream));
        c->esrKeyA.release(); c->esrKeyB.release(); c->esrValA.release(); c->esrValB.release();
    }
    *listOut = c->esrShort.as<uint32_t>(); *nOut = nShort;
    return KASA_OK;
}

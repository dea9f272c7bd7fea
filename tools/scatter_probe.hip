// scatter_probe.hip -- measures what a random permutation of fixed-size records costs on this GPU: the step that
// moves per-query results from sorted-k-mer order to read order (group_kernel's record stores).  Writes N records of R
// bytes to random slots (slot = bijective hash of the index) and gathers them back, for R = 4 ... 64, by record size,
// footprint (window), resident wavefronts, workgroup size, store flavour.  Prints text and, with a second argument, a
// JSON file (profiles/rNN_scatter_probe.json) that bench.py reads for the "scatter" bound of group_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/scatter_probe tools/scatter_probe.hip && tools/scatter_probe 30 out.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask)   // bijection on [0, 2^b): odd multiply + xorshift
{
    i = (i * 0x9E3779B1u) & mask;
    i ^= i >> 15; i = (i * 0x85EBCA6Bu) & mask; i ^= i >> 13;           // xorshifts and odd multiplies are invertible mod 2^b
    return i & mask;
}

template <int WORDS>   // record = WORDS x 16 bytes (WORDS = 0: 8-byte record, WORDS = -1: 4-byte record)
__global__ void scatter_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    if constexpr (WORDS == -1) reinterpret_cast<uint32_t *>(dst)[s] = i;
    else if constexpr (WORDS == 0) reinterpret_cast<uint2 *>(dst)[s] = make_uint2(i, s);
    else {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) dst[(size_t)s * WORDS + w] = make_uint4(i, s, w, 0);
    }
}

// 32-byte records written by PAIRS of lanes: both halves of a record leave in one store instruction
__global__ void scatter_pair_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = t >> 1;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    dst[(size_t)s * 2 + (t & 1u)] = make_uint4(i, s, t & 1u, 0);
}

// 32-byte records whose slots stay inside a WINDOW of 2^wbits records that moves with the index: what the scatter costs when the
// batch is taken in pieces (the footprint the stores of one moment touch is the window, not the whole buffer)
__global__ void scatter_window_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t wmask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = (i & ~wmask) | perm(i & wmask, wmask);
    dst[(size_t)s * 2] = make_uint4(i, s, 0, 0);
    dst[(size_t)s * 2 + 1] = make_uint4(i, s, 1, 0);
}

// 32-byte records with non-temporal stores (the records are read once, by another kernel, much later)
__global__ void scatter_nt_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 a = {i, s, 0u, 0u}, b = {i, s, 1u, 0u};
    __builtin_nontemporal_store(a, reinterpret_cast<v4 *>(dst + (size_t)s * 2));
    __builtin_nontemporal_store(b, reinterpret_cast<v4 *>(dst + (size_t)s * 2 + 1));
}

// runs of RUN consecutive 32-byte records share a random place (what binning records by destination block would buy)
template <int RUN>
__global__ void scatter_run_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i / RUN, mask / RUN) * RUN + i % RUN;
    dst[(size_t)s * 2] = make_uint4(i, s, 0, 0);
    dst[(size_t)s * 2 + 1] = make_uint4(i, s, 1, 0);
}

// 64-byte records written by QUADS of lanes (one store instruction per record, 16 B per lane) or PAIRS (two instructions)
template <int LANES>
__global__ void scatter64_team_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / LANES), part = (uint32_t)(t % LANES);
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
#pragma unroll
    for (int w = 0; w < 4 / LANES; ++w) dst[(size_t)s * 4 + part * (4 / LANES) + w] = make_uint4(i, s, part, w);
}
// a 32-byte payload in a 64-byte cell: the payload alone (a partial write of the cell) or payload + padding by a lane pair
__global__ void scatter32in64_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    dst[(size_t)s * 4] = make_uint4(i, s, 0, 0);
    dst[(size_t)s * 4 + 1] = make_uint4(i, s, 1, 0);
}
// the record leaves through LDS: a wavefront's 64 records of 64 bytes are written as 4 instructions of 16 whole records
// (what group_kernel would do: one lane owns a record, four lanes store it)
__global__ void scatter64_lds_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    __shared__ uint4 sh[256 * 4];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = i < n ? perm(i, mask) : 0xFFFFFFFFu;
    uint4 *mine = sh + wv * 256;
#pragma unroll
    for (int w = 0; w < 4; ++w) mine[lane * 4 + w] = make_uint4(i, s, w, 0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sk = __shfl(s, 16 * k + (lane >> 2));
        if (sk != 0xFFFFFFFFu) dst[(size_t)sk * 4 + (lane & 3)] = mine[k * 64 + lane];
    }
}


// ROUND 6: a 32-byte payload written as a FULL 64-byte cell -- the payload and 32 bytes of anything else -- by lane quads
// through LDS (one lane owns a record, four lanes store its cell in ONE instruction: 16 whole cells per instruction)
__global__ void scatter32as64_lds_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    __shared__ uint4 sh[256 * 2];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = i < n ? perm(i, mask) : 0xFFFFFFFFu;
    uint4 *mine = sh + wv * 128;
    mine[lane * 2] = make_uint4(i, s, 0, 0); mine[lane * 2 + 1] = make_uint4(i, s, 1, 0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sk = __shfl(s, 16 * k + (lane >> 2));
        // lanes 0, 1 of a quad carry the payload, lanes 2, 3 the filler (here: the payload once more)
        if (sk != 0xFFFFFFFFu) dst[(size_t)sk * 4 + (lane & 3)] = mine[(16 * k + (lane >> 2)) * 2 + (lane & 1)];
    }
}
// the same, the filler being zeros made in registers (no second LDS read)
__global__ void scatter32as64_zero_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    __shared__ uint4 sh[256 * 2];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = i < n ? perm(i, mask) : 0xFFFFFFFFu;
    uint4 *mine = sh + wv * 128;
    mine[lane * 2] = make_uint4(i, s, 0, 0); mine[lane * 2 + 1] = make_uint4(i, s, 1, 0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sk = __shfl(s, 16 * k + (lane >> 2));
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((lane & 2) == 0) v = mine[(16 * k + (lane >> 2)) * 2 + (lane & 1)];
        if (sk != 0xFFFFFFFFu) dst[(size_t)sk * 4 + (lane & 3)] = v;
    }
}
// 32-byte records in 32-byte slots, but stored by lane PAIRS through LDS (one instruction = 32 records of 32 bytes, each a
// contiguous half cell): is it the number of instructions or the partial cell that costs?
__global__ void scatter32_pairs_lds_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    __shared__ uint4 sh[256 * 2];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t s = i < n ? perm(i, mask) : 0xFFFFFFFFu;
    uint4 *mine = sh + wv * 128;
    mine[lane * 2] = make_uint4(i, s, 0, 0); mine[lane * 2 + 1] = make_uint4(i, s, 1, 0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t sk = __shfl(s, 32 * k + (lane >> 1));
        if (sk != 0xFFFFFFFFu) dst[(size_t)sk * 2 + (lane & 1)] = mine[k * 64 + lane];
    }
}
// what the READERS of such cells pay: a wavefront streams 64 consecutive cells and wants the first 32 bytes of each
// (HALF = 1: loads the payload halves only; HALF = 0: loads whole cells), 64-byte cells; and 32-byte records back to back
template <int CELL_WORDS, int READ_WORDS>
__global__ void stream_read_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < READ_WORDS; ++w) acc += src[(size_t)i * CELL_WORDS + w].x;
    if (acc == 0x12345678u) out[i] = acc;
}
// the score kernels' pattern: a LANE streams its own run of RUN consecutive cells (a read's queries), a wavefront = 64 runs
template <int CELL_WORDS, int READ_WORDS, int RUN>
__global__ void lane_stream_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ out, uint32_t nRuns)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRuns) return;
    uint32_t acc = 0;
    const uint4 *p = src + (size_t)r * RUN * CELL_WORDS;
    for (int j = 0; j < RUN; ++j) {
#pragma unroll
        for (int w = 0; w < READ_WORDS; ++w) acc += p[(size_t)j * CELL_WORDS + w].x;
    }
    if (acc == 0x12345678u) out[r] = acc;
}

template <int WORDS>
__global__ void gather_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ out, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    uint32_t acc = 0;
    if constexpr (WORDS == -1) acc = reinterpret_cast<const uint32_t *>(src)[s];
    else if constexpr (WORDS == 0) acc = reinterpret_cast<const uint2 *>(src)[s].x;
    else {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) acc += src[(size_t)s * WORDS + w].x;
    }
    out[i] = acc;
}

// streaming stores of the same bytes: the rate the chip reaches when the slots are NOT permuted
__global__ void stream_kernel(uint4 *__restrict__ dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[(size_t)i * 2] = make_uint4(i, i, 0, 0);
    dst[(size_t)i * 2 + 1] = make_uint4(i, i, 1, 0);
}

struct Row { std::string what; int bytes; double ms, grec; };
static std::vector<Row> rows;

template <class F> static double timeit(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

static void note(const std::string &what, int bytes, double ms, uint32_t n)
{
    rows.push_back({what, bytes, ms, n / ms / 1e6});
    printf("%-78s %3d B  %8.2f ms  %6.2f G rec/s  %7.1f GB/s\n", what.c_str(), bytes, ms, n / ms / 1e6, (double)n * bytes / ms / 1e6);
    fflush(stdout);
}

template <int WORDS> static void run(uint4 *buf, uint32_t *out, uint32_t n, uint32_t mask)
{
    const int bytes = WORDS > 0 ? WORDS * 16 : (WORDS == 0 ? 8 : 4);
    note("scatter, whole buffer, 256-thread workgroups", bytes, timeit([&] { scatter_kernel<WORDS><<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("gather, whole buffer, 256-thread workgroups", bytes, timeit([&] { gather_kernel<WORDS><<<(n + 255) / 256, 256>>>(buf, out, n, mask); }), n);
}

int main(int argc, char **argv)
{
    int bits = argc > 1 ? atoi(argv[1]) : 30;            // 2^30 records ~ the 1.3e9 queries of a 10 M-read batch
    const char *json = argc > 2 ? argv[2] : nullptr;
    const uint32_t n = 1u << bits, mask = n - 1;
    uint4 *buf; uint32_t *out;
    if (hipMalloc(&buf, (size_t)n * 64) != hipSuccess || hipMalloc(&out, (size_t)n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, (size_t)n * 64);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("%s, %d CUs; %u records\n", prop.name, prop.multiProcessorCount, n);
    run<-1>(buf, out, n, mask); run<0>(buf, out, n, mask); run<1>(buf, out, n, mask); run<2>(buf, out, n, mask); run<3>(buf, out, n, mask); run<4>(buf, out, n, mask);
    note("streaming stores (slot = index): the unpermuted rate", 32, timeit([&] { stream_kernel<<<(n + 255) / 256, 256>>>(buf, n); }), n);
    note("scatter by lane pairs (both halves of a record in one store instruction)", 32,
         timeit([&] { scatter_pair_kernel<<<(unsigned)(((uint64_t)n * 2 + 255) / 256), 256>>>(buf, n, mask); }), n);
    note("scatter of 64-byte records by lane pairs (two store instructions)", 64,
         timeit([&] { scatter64_team_kernel<2><<<(unsigned)(((uint64_t)n * 2 + 255) / 256), 256>>>(buf, n, mask); }), n);
    note("scatter of 64-byte records, lane per record, stored by quads through LDS (4 instructions of 16 records)", 64,
         timeit([&] { scatter64_lds_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("scatter of a 32-byte payload into 64-byte cells (half of each cell written)", 32,
         timeit([&] { scatter32in64_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("scatter, non-temporal stores", 32, timeit([&] { scatter_nt_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    for (int threads : {64, 128, 512, 1024})
        note("scatter, workgroups of " + std::to_string(threads) + " threads", 32,
             timeit([&] { scatter_kernel<2><<<(n + threads - 1) / threads, threads>>>(buf, n, mask); }), n);
    // fewer resident wavefronts (dynamic LDS as ballast): does the scatter rate depend on how many stores are in flight?
    hipFuncSetAttribute((const void *)scatter_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds : {16, 32, 64, 128}) {
        const int wgs = 160 / lds > 8 ? 8 : 160 / lds;
        note("scatter, " + std::to_string(wgs * 4) + " wavefronts per CU (" + std::to_string(lds) + " KB of LDS per 256-thread workgroup)", 32,
             timeit([&] { scatter_kernel<2><<<(n + 255) / 256, 256, lds * 1024>>>(buf, n, mask); }), n);
    }
    for (int wb = 16; wb <= bits; wb += 2) {
        char what[128];
        snprintf(what, sizeof what, "scatter, slots within windows of 2^%d records (%.1f MB)", wb, (double)(1u << wb) * 32 / 1e6);
        note(what, 32, timeit([&] { scatter_window_kernel<<<(n + 255) / 256, 256>>>(buf, n, (1u << wb) - 1u); }), n);
    }
    note("scatter, runs of 2 records (64 B) share a random place", 32, timeit([&] { scatter_run_kernel<2><<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("scatter, runs of 4 records (128 B) share a random place", 32, timeit([&] { scatter_run_kernel<4><<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("scatter, runs of 8 records (256 B) share a random place", 32, timeit([&] { scatter_run_kernel<8><<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("scatter, runs of 16 records (512 B) share a random place", 32, timeit([&] { scatter_run_kernel<16><<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("ROUND 6: 32-byte payload written as a FULL 64-byte cell by lane quads through LDS (payload + 32 B of anything)", 32,
         timeit([&] { scatter32as64_lds_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("ROUND 6: the same, the other 32 bytes zeros from registers", 32,
         timeit([&] { scatter32as64_zero_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("ROUND 6: 32-byte records in 32-byte slots stored by lane pairs through LDS (2 instructions of 32 half cells)", 32,
         timeit([&] { scatter32_pairs_lds_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); }), n);
    note("ROUND 6 readers: wavefront streams consecutive 32-byte records (2 x 16 B per lane)", 32,
         timeit([&] { stream_read_kernel<2, 2><<<(n + 255) / 256, 256>>>(buf, out, n); }), n);
    note("ROUND 6 readers: wavefront streams 64-byte cells, loads the first 32 bytes of each", 32,
         timeit([&] { stream_read_kernel<4, 2><<<(n + 255) / 256, 256>>>(buf, out, n); }), n);
    note("ROUND 6 readers: wavefront streams 64-byte cells, loads all 64 bytes", 64,
         timeit([&] { stream_read_kernel<4, 4><<<(n + 255) / 256, 256>>>(buf, out, n); }), n);
    {
        const uint32_t runs = n / 128;
        note("ROUND 6 readers: a lane streams its own 128 consecutive 32-byte records (score_main's pattern)", 32,
             timeit([&] { lane_stream_kernel<2, 2, 128><<<(runs + 255) / 256, 256>>>(buf, out, runs); }), runs * 128);
        note("ROUND 6 readers: a lane streams its own 128 consecutive 64-byte cells, first 32 bytes of each", 32,
             timeit([&] { lane_stream_kernel<4, 2, 128><<<(runs + 255) / 256, 256>>>(buf, out, runs); }), runs * 128);
        note("ROUND 6 readers: a lane streams its own 128 consecutive 64-byte cells, all 64 bytes", 64,
             timeit([&] { lane_stream_kernel<4, 4, 128><<<(runs + 255) / 256, 256>>>(buf, out, runs); }), runs * 128);
    }
    if (json) {
        FILE *f = fopen(json, "w");
        if (!f) { perror(json); return 1; }
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"records\": %u, \"timing\": \"best of 3 launches after one warm-up, HIP events\",\n \"rows\": [\n", prop.name, prop.multiProcessorCount, n);
        for (size_t k = 0; k < rows.size(); ++k)
            fprintf(f, "  {\"what\": \"%s\", \"record_bytes\": %d, \"ms\": %.3f, \"g_records_per_s\": %.3f, \"gb_per_s\": %.1f}%s\n", rows[k].what.c_str(), rows[k].bytes,
                    rows[k].ms, rows[k].grec, rows[k].grec * rows[k].bytes, k + 1 < rows.size() ? "," : "");
        fprintf(f, " ]}\n");
        fclose(f);
    }
    return 0;
}

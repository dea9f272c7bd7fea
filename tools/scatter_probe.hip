// scatter_probe.hip -- measures what a random permutation of fixed-size records costs on this GPU: the step that
// moves per-query results from sorted-k-mer order to read order.  Writes N records of R bytes to random slots
// (slot = bijective hash of the index), and gathers them back, for R = 8, 16, 32, 48, 64.
//   hipcc --offload-arch=gfx950 -O3 -o scatter_probe tools/scatter_probe.hip && ./scatter_probe [N]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask)   // bijection on [0, 2^b): odd multiply + xorshift
{
    i = (i * 0x9E3779B1u) & mask;
    i ^= i >> 15; i = (i * 0x85EBCA6Bu) & mask; i ^= i >> 13;           // xorshifts and odd multiplies are invertible mod 2^b
    return i & mask;
}

template <int WORDS>   // record = WORDS x 16 bytes (WORDS = 0: 8-byte record)
__global__ void scatter_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    if constexpr (WORDS == 0) reinterpret_cast<uint2 *>(dst)[s] = make_uint2(i, s);
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

template <int WORDS>
__global__ void gather_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ out, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
    uint32_t acc = 0;
    if constexpr (WORDS == 0) acc = reinterpret_cast<const uint2 *>(src)[s].x;
    else {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) acc += src[(size_t)s * WORDS + w].x;
    }
    out[i] = acc;
}

template <int WORDS> static void run(uint4 *buf, uint32_t *out, uint32_t n, uint32_t mask)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int bytes = WORDS ? WORDS * 16 : 8;
    float msS = 0, msG = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a); scatter_kernel<WORDS><<<(n + 255) / 256, 256>>>(buf, n, mask); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&msS, a, b);
        hipEventRecord(a); gather_kernel<WORDS><<<(n + 255) / 256, 256>>>(buf, out, n, mask); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&msG, a, b);
    }
    printf("record %2d B: scatter %7.2f ms (%6.1f GB/s, %5.2f G rec/s)   gather %7.2f ms (%6.1f GB/s)\n", bytes, msS,
           (double)n * bytes / msS / 1e6, n / msS / 1e6, msG, (double)n * bytes / msG / 1e6);
}

int main(int argc, char **argv)
{
    int bits = argc > 1 ? atoi(argv[1]) : 30;            // 2^30 records ~ the 1.3e9 queries of a 10 M-read batch
    const uint32_t n = 1u << bits, mask = n - 1;
    uint4 *buf; uint32_t *out;
    if (hipMalloc(&buf, (size_t)n * 64) != hipSuccess || hipMalloc(&out, (size_t)n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, (size_t)n * 64);
    printf("%u records\n", n);
    run<0>(buf, out, n, mask); run<1>(buf, out, n, mask); run<2>(buf, out, n, mask); run<3>(buf, out, n, mask); run<4>(buf, out, n, mask);
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a); scatter_pair_kernel<<<(unsigned)(((uint64_t)n * 2 + 255) / 256), 256>>>(buf, n, mask); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("record 32 B by lane pairs: scatter %7.2f ms (%5.2f G rec/s)\n", ms, n / ms / 1e6);
    }
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a); scatter_nt_kernel<<<(n + 255) / 256, 256>>>(buf, n, mask); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("record 32 B, non-temporal stores: scatter %7.2f ms (%5.2f G rec/s)\n", ms, n / ms / 1e6);
        for (int threads : {64, 128, 512, 1024}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a); scatter_kernel<2><<<(n + threads - 1) / threads, threads>>>(buf, n, mask); hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            printf("record 32 B, workgroups of %4d: scatter %7.2f ms (%5.2f G rec/s)\n", threads, ms, n / ms / 1e6);
        }
    }
    {   // fewer resident wavefronts (dynamic LDS as ballast): does the scatter rate depend on how many stores are in flight?
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        hipFuncSetAttribute((const void *)scatter_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int lds : {0, 16 * 1024, 32 * 1024, 64 * 1024, 128 * 1024}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a); scatter_kernel<2><<<(n + 255) / 256, 256, lds>>>(buf, n, mask); hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            printf("record 32 B, %3d KB of LDS per 256-thread workgroup: scatter %7.2f ms (%5.2f G rec/s)\n", lds / 1024, ms, n / ms / 1e6);
        }
    }
    for (int wb = 16; wb <= bits; wb += 6) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a); scatter_window_kernel<<<(n + 255) / 256, 256>>>(buf, n, (1u << wb) - 1u); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("record 32 B, slots within windows of 2^%d records (%7.1f MB): scatter %7.2f ms (%5.2f G rec/s)\n", wb, (double)(1u << wb) * 32 / 1e6, ms, n / ms / 1e6);
    }
    return 0;
}

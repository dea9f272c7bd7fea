// sort_probe.hip -- how fast does rocPRIM sort 1.3e9 (60-bit key, 32-bit payload) pairs with wider digits?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sort_probe tools/sort_probe.hip && /tmp/sort_probe
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <unsigned BITS, unsigned BLK, unsigned IPT>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BLK, IPT>, BITS,
                                                                           rocprim::block_radix_rank_algorithm::match>, 1024 * 1024>;
__global__ void fill(uint64_t *k, uint32_t *v, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        k[i] = x & ((1ull << 60) - 1); v[i] = (uint32_t)i;
    }
}
__global__ void check(const uint64_t *k, size_t n, unsigned *bad)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x + 1; i < n; i += (size_t)gridDim.x * blockDim.x) if (k[i - 1] > k[i]) atomicAdd(bad, 1u);
}
template <class C> int one(const char *name, uint64_t *a, uint64_t *b, uint32_t *va, uint32_t *vb, size_t n, unsigned *bad)
{
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs<C>(nullptr, bytes, a, b, va, vb, n, 0u, 60u, 0));
    void *tmp; CK(hipMalloc(&tmp, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        CK(rocprim::radix_sort_pairs<C>(tmp, bytes, a, b, va, vb, n, 0u, 60u, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipMemset(bad, 0, 4));
    check<<<4096, 256>>>(b, n, bad);
    unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    printf("%-28s %8.2f ms  tmp %.1f GB  unsorted pairs %u\n", name, best, bytes / 1e9, h);
    CK(hipFree(tmp));
    return 0;
}
int main()
{
    const size_t n = 1300000000;
    uint64_t *a, *b; uint32_t *va, *vb; unsigned *bad;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4)); CK(hipMalloc(&bad, 4));
    fill<<<4096, 256>>>(a, va, n);
    if (one<rocprim::default_config>("default", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<8, 512, 16>>("8 bits match 512x16", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<8, 512, 20>>("8 bits match 512x20", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<8, 1024, 8>>("8 bits match 1024x8", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<8, 1024, 10>>("8 bits match 1024x10", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<8, 256, 24>>("8 bits match 256x24", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<8, 256, 32>>("8 bits match 256x32", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<9, 512, 20>>("9 bits match 512x20", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<9, 1024, 10>>("9 bits match 1024x10", a, b, va, vb, n, bad)) return 1;
    if (one<Cfg<10, 1024, 10>>("10 bits match 1024x10", a, b, va, vb, n, bad)) return 1;
    return 0;
}

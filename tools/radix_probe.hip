// radix_probe.hip -- the hand-written radix pass of the query sort against rocPRIM's onesweep, stand-alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/radix_probe tools/radix_probe.hip && tools/radix_probe [n]
// Keys: 12 letters of 5 bits from a skewed alphabet (as amino-acid-like k-mers are), payload = index.  Sorts the top 40 key
// bits in 5 stable passes of 8 bits both ways and compares the results element by element.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#include "../kasa_amd/csrc/kasa_radix.h"

__global__ void fill(uint64_t *k, uint32_t *v, size_t n)
{
    // letters with the frequencies of codon degeneracy: 6,6,6,4,4,4,4,4,3,2 x9,1,1 (of 64) -- skewed first letters
    const uint8_t deg[21] = {6, 6, 6, 4, 4, 4, 4, 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 3};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull, key = 0;
        for (int l = 0; l < 12; ++l) {
            x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
            int r = (int)(x % 64), letter = 0;
            while (r >= deg[letter]) { r -= deg[letter]; ++letter; }
            key = (key << 5) | (uint64_t)(letter + 1);
        }
        k[i] = key; v[i] = (uint32_t)i;
    }
}
__global__ void differ(const uint64_t *a, const uint64_t *b, const uint32_t *va, const uint32_t *vb, size_t n, unsigned *bad)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != b[i] || va[i] != vb[i]) atomicAdd(bad, 1u);
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1300000000ull;
    uint64_t *a, *b, *c, *d; uint32_t *va, *vb, *vc, *vd; unsigned *bad;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&d, n * 8));
    CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4)); CK(hipMalloc(&vc, n * 4)); CK(hipMalloc(&vd, n * 4)); CK(hipMalloc(&bad, 4));
    fill<<<4096, 256>>>(a, va, n);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // rocPRIM: bits 20 .. 60
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, bytes, a, b, va, vb, n, 20u, 60u, 0));
    void *tmp; CK(hipMalloc(&tmp, bytes));
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        CK(rocprim::radix_sort_pairs(tmp, bytes, a, b, va, vb, n, 20u, 60u, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("n = %zu\nrocPRIM   bits 20..60: %8.2f ms (tmp %.1f GB)\n", n, best, bytes / 1e9);
    // ours: same bits; the input buffer is overwritten (ping-pong), so every round starts from a fresh copy of `a` in c
    size_t sbytes = kasa_radix::scratch_bytes<uint64_t>(n);
    void *scratch; CK(hipMalloc(&scratch, sbytes));
    for (int variant = 0; variant < 3; ++variant) {
        best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            CK(hipMemcpy(c, a, n * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(vc, va, n * 4, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(e0));
            uint64_t *ko; uint32_t *vo;
            CK(kasa_radix::sort_pairs<uint64_t>(c, vc, d, vd, (uint32_t)n, 20, 40, scratch, 0, &ko, &vo, variant));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            if (it == 0) {
                CK(hipMemset(bad, 0, 4));
                differ<<<4096, 256>>>(b, ko, vb, vo, n, bad);
                unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
                printf("kasa_radix (variant %d) differs from rocPRIM in %u places\n", variant, h);
            }
        }
        printf("kasa_radix bits 20..60, variant %d (tiles per XCD run: %d): %8.2f ms (scratch %.2f GB)\n", variant, variant == 0 ? 0 : (variant == 1 ? 8 : 32), best, sbytes / 1e9);
    }
    return 0;
}

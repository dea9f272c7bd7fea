// Does the random-store rate of 32-byte records depend on WHERE hipMalloc put the buffer?  (The group stage's kernels measure
// 56 or 61 ms on the same batch in one process, before and after the record buffer was freed and allocated again: DESIGN.md 3c.)
// One process, 2^30 records of 32 bytes (34 GB) scattered to random slots of buffers obtained in different ways.
//   hipcc --offload-arch=gfx950 -O3 -o tools/place_probe tools/place_probe.hip && tools/place_probe [bits] [out.json]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask)   // bijection on [0, 2^b): odd multiplies and xorshifts
{
    i = (i * 0x9E3779B1u) & mask;
    i ^= i >> 15; i = (i * 0x85EBCA6Bu) & mask; i ^= i >> 13;
    return i & mask;
}

template <int WORDS>
__global__ void scatter_kernel(uint4 *__restrict__ dst, uint32_t n, uint32_t mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm(i, mask);
#pragma unroll
    for (int w = 0; w < WORDS; ++w) dst[(size_t)s * WORDS + w] = make_uint4(i, s, w, 0);
}

struct Row { std::string what; double ms32, ms64, ms16; unsigned long long addr; };
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

static uint32_t N, MASK;

static void measure(const std::string &what, void *base)
{
    uint4 *buf = (uint4 *)base;
    const double a = timeit([&] { scatter_kernel<2><<<(N + 255) / 256, 256>>>(buf, N, MASK); });
    const double c = timeit([&] { scatter_kernel<1><<<(N + 255) / 256, 256>>>(buf, N, MASK); });
    rows.push_back({what, a, 0, c, (unsigned long long)(uintptr_t)base});
    printf("%-86s at %#14llx  32 B: %6.2f ms %6.2f G rec/s   16 B: %6.2f G rec/s\n", what.c_str(), (unsigned long long)(uintptr_t)base, a, N / a / 1e6, N / c / 1e6);
    fflush(stdout);
}

static void *alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc(%zu) failed\n", bytes); (void)hipGetLastError(); return nullptr; }
    return p;
}

int main(int argc, char **argv)
{
    const int bits = argc > 1 ? atoi(argv[1]) : 30;
    const char *json = argc > 2 ? argv[2] : nullptr;
    N = 1u << bits; MASK = N - 1;
    const size_t B = (size_t)N * 32;
    size_t freeB = 0, totalB = 0;
    hipMemGetInfo(&freeB, &totalB);
    printf("free %.1f GB of %.1f GB\n", freeB / 1e9, totalB / 1e9);

    if (argc > 3 && std::string(argv[3]) == "vmm") {
        // ONE piece of physical memory (hipMemCreate) mapped at different virtual addresses: does the rate follow the address?
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        const size_t G = 1ull << 30, M2 = 2ull << 20;
        const size_t sz = (B + M2 - 1) / M2 * M2;
        for (int handle = 0; handle < 3; ++handle) {
            hipMemGenericAllocationHandle_t h, hold;
            if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); break; }
            void *va = nullptr;
            if (hipMemAddressReserve(&va, sz + 2 * G, 0, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve failed\n"); break; }
            const uintptr_t base = ((uintptr_t)va + G - 1) / G * G;       // 1 GB-aligned inside the reservation
            printf("handle %d: reservation at %#llx, 1 GB-aligned base %#llx\n", handle, (unsigned long long)(uintptr_t)va, (unsigned long long)base);
            const size_t deltas[] = {0, M2, 2 * M2, 4 * M2, 8 * M2, 16 * M2, 32 * M2, 64 * M2, 128 * M2, 256 * M2, 3 * M2, 0};
            for (size_t d : deltas) {
                void *at = (void *)(base + d);
                if (hipMemMap(at, sz, 0, h, 0) != hipSuccess) { printf("hipMemMap failed\n"); break; }
                hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                if (hipMemSetAccess(at, sz, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); break; }
                measure("vmm: handle " + std::to_string(handle) + " mapped at 1 GB-aligned base + " + std::to_string(d >> 20) + " MB", at);
                hipDeviceSynchronize();
                hipMemUnmap(at, sz);
            }
            hipMemAddressFree(va, sz + 2 * G);
            // keep this physical piece away from the next handle: hold a second handle of the same size while the first goes back
            if (hipMemCreate(&hold, sz, &prop, 0) == hipSuccess) { hipMemRelease(h); h = hold; }
            hipMemRelease(h);
        }
        goto done;
    }
    if (argc > 3 && std::string(argv[3]) == "map") {
        // the whole memory in pieces that are held together: is the rate a property of the place?
        std::vector<void *> big;
        for (int i = 0; i < 8; ++i) { void *q = alloc(B); if (!q) break; big.push_back(q); }
        for (size_t i = 0; i < big.size(); ++i) measure("map: 34 GB piece " + std::to_string(i) + " of " + std::to_string(big.size()) + " held together", big[i]);
        for (size_t i = big.size(); i-- > 0;) measure("map: the same pieces again, backwards: piece " + std::to_string(i), big[i]);
        for (void *q : big) hipFree(q);
        const uint32_t n0 = N;
        N = 1u << 27; MASK = N - 1;                                  // 4 GB pieces
        std::vector<void *> small;
        for (int i = 0; i < 68; ++i) { void *q = alloc((size_t)N * 32); if (!q) break; small.push_back(q); }
        for (size_t i = 0; i < small.size(); ++i) measure("map: 4 GB piece " + std::to_string(i), small[i]);
        for (void *q : small) hipFree(q);
        N = n0; MASK = N - 1;
        goto done;
    }
    {
    void *p = alloc(B);
    if (!p) return 1;
    measure("1. the first allocation of the process, exactly 34 GB", p);
    hipFree(p);

    p = alloc(B);
    measure("2. freed and allocated again, the same size", p);
    hipFree(p);

    p = alloc(B + B / 16 + 256);                                     // (the library's grow-only buffers ask for 1/16 more + 256 bytes)
    measure("3. 34 GB + 1/16 + 256 bytes (a size that is no multiple of anything)", p);
    hipFree(p);

    p = alloc(2 * B);
    measure("4. 68 GB, its first half", p);
    measure("5. the same 68 GB, its second half", (char *)p + B);
    measure("6. the same 68 GB, from 1 GB + 4 KB in", (char *)p + (1ull << 30) + 4096);
    hipFree(p);

    {   // a history like the library's first steps: buffers grow (free + allocate larger), small ones stay in between
        std::vector<void *> keep;
        for (int i = 0; i < 40; ++i) {
            void *a = alloc((size_t)(50 + 37 * i) * 1000 * 1000 + 256 * i);
            void *b = alloc((size_t)(900 + 211 * i) * 1000 * 1000);
            if (a) keep.push_back(a);
            if (b) hipFree(b);
        }
        void *big = alloc(5ull * 1000 * 1000 * 1000 + 4096);         // the index
        p = alloc(B + B / 16 + 256);
        measure("7. after 40 small buffers that stay, 40 larger ones freed again, a 5 GB one: 34 GB + 1/16", p);
        void *q = alloc(2 * B + B / 8 + 256);
        if (q) { measure("8. then 68 GB + 1/8 beside it, its first half (what the record buffer is after a step with 64-byte cells)", q); }
        hipFree(p);
        if (q) { measure("9. the same after the 34 GB one was freed", q); hipFree(q); }
        p = alloc(B + B / 16 + 256);
        measure("10. 34 GB + 1/16 allocated again into that history", p);
        hipFree(p);
        for (void *a : keep) hipFree(a);
        hipFree(big);
    }

    {   // the virtual-memory API: physical memory by explicit handles, mapped at an address of our choice of alignment
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gMin = 0, gRec = 0;
        hipMemGetAllocationGranularity(&gMin, &prop, hipMemAllocationGranularityMinimum);
        hipMemGetAllocationGranularity(&gRec, &prop, hipMemAllocationGranularityRecommended);
        printf("allocation granularity: minimum %zu, recommended %zu bytes\n", gMin, gRec);
        const size_t gran = gRec ? gRec : (2u << 20);
        const size_t sz = (B + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h;
        void *va = nullptr;
        if (hipMemCreate(&h, sz, &prop, 0) == hipSuccess && hipMemAddressReserve(&va, sz, 1ull << 30, nullptr, 0) == hipSuccess && hipMemMap(va, sz, 0, h, 0) == hipSuccess) {
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            if (hipMemSetAccess(va, sz, &acc, 1) == hipSuccess) measure("11. hipMemCreate (one handle) mapped at a 1 GB-aligned address", va);
            else printf("hipMemSetAccess failed\n");
            hipMemUnmap(va, sz); hipMemAddressFree(va, sz); hipMemRelease(h);
        } else { printf("virtual-memory API: not available here\n"); (void)hipGetLastError(); }
    }

    p = alloc(B);
    measure("12. the plain allocation of row 1 once more, at the end", p);
    hipFree(p);
    }
done:

    if (json) {
        FILE *f = fopen(json, "w");
        if (f) {
            fprintf(f, "{\"records\": %u, \"record_bytes\": 32, \"timing\": \"best of 3 launches after one warm-up, HIP events\", \"rows\": [", N);
            for (size_t i = 0; i < rows.size(); ++i)
                fprintf(f, "%s{\"what\": \"%s\", \"address\": \"%#llx\", \"ms\": %.3f, \"g_records_per_s\": %.3f, \"g_records_per_s_16_byte_records\": %.3f}", i ? ", " : "",
                        rows[i].what.c_str(), rows[i].addr, rows[i].ms32, N / rows[i].ms32 / 1e6, N / rows[i].ms16 / 1e6);
            fprintf(f, "]}\n");
            fclose(f);
        }
    }
    return 0;
}

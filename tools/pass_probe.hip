// pass_probe.hip -- the query sort's radix passes (kasa_amd/csrc/kasa_radix.h) alone, on random 60-bit keys, per measurement
// tap of pass_kernel: what the look-back between tiles and the order of the payload loads cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pass_probe.hip -o tools/pass_probe
//   tools/pass_probe [million pairs, default 1150] [out.json]
// For every mode: five passes over the top 40 of 60 key bits (the product runs four over 1.15e9 pairs for a 10 M-read batch),
// HIP-event time of three repetitions, and -- for the modes that sort -- a check that the result is ordered by those bits
// and is a permutation (sum and xor of key ^ payload-hash).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include "../kasa_amd/csrc/kasa_radix.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__global__ void fill_kernel(uint64_t *k, uint32_t *v, uint32_t n, uint64_t seed)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        k[i] = mix(i + seed) >> 4;                                       // 60 bits
        v[i] = (uint32_t)i;
    }
}
// ordered by the top 40 of 60 bits?  + sum / xor over (key + mix(payload)): the same multiset before and after
__global__ void check_kernel(const uint64_t *k, const uint32_t *v, uint32_t n, unsigned long long *out)
{
    unsigned long long bad = 0, sum = 0, x = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i && (k[i - 1] >> 20) > (k[i] >> 20)) ++bad;
        if (i && (k[i - 1] >> 20) == (k[i] >> 20) && v[i - 1] > v[i]) ++bad;   // stable: payloads were 0, 1, 2, ...
        const unsigned long long h = k[i] + mix(v[i]);
        sum += h; x ^= h;
    }
    atomicAdd(&out[0], bad); atomicAdd(&out[1], sum); atomicXor(&out[2], x);
}

int main(int argc, char **argv)
{
    const uint32_t n = (uint32_t)((argc > 1 ? atof(argv[1]) : 1150.0) * 1e6);
    const char *outPath = argc > 2 ? argv[2] : nullptr;
    uint64_t *kA, *kB; uint32_t *vA, *vB; void *scratch; unsigned long long *chk;
    CHECK(hipMalloc(&kA, (size_t)n * 8)); CHECK(hipMalloc(&kB, (size_t)n * 8));
    CHECK(hipMalloc(&vA, (size_t)n * 4)); CHECK(hipMalloc(&vB, (size_t)n * 4));
    CHECK(hipMalloc(&scratch, kasa_radix::scratch_bytes<uint64_t>(n)));
    CHECK(hipMalloc(&chk, 64));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Mode { int variant; const char *name; bool sorts; };
    using namespace kasa_radix;
    const Mode modes[] = {{0, "product (look-back after the keys are ordered in LDS)", true}, {MODE_FIRST, "look-back first (rounds 2-3)", true},
                          {MODE_NOLOOK, "no look-back (timing tap)", false}, {0, "product (again)", true}};
    unsigned long long ref[3] = {0, 0, 0};
    std::string json = "{\"pairs\": " + std::to_string(n) + ", \"passes\": 5, \"rows\": [";
    bool first = true, allOk = true;
    for (const Mode &m : modes) {
        float best = 1e30f, sumMs = 0.0f;
        bool ok = true;
        for (int rep = 0; rep < 3; ++rep) {
            fill_kernel<<<4096, 256, 0, st>>>(kA, vA, n, 12345);
            if (ref[1] == 0) {
                CHECK(hipMemsetAsync(chk, 0, 64, st));
                check_kernel<<<4096, 256, 0, st>>>(kA, vA, n, chk);
                unsigned long long h[3]; CHECK(hipMemcpyAsync(h, chk, 24, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
                ref[1] = h[1]; ref[2] = h[2];
            }
            uint64_t *kRes; uint32_t *vRes;
            CHECK(hipEventRecord(e0, st));
            CHECK(kasa_radix::sort_pairs<uint64_t>(kA, vA, kB, vB, n, 20, 40, scratch, st, &kRes, &vRes, m.variant));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sumMs += ms;
            if (m.sorts && rep == 0) {
                CHECK(hipMemsetAsync(chk, 0, 64, st));
                check_kernel<<<4096, 256, 0, st>>>(kRes, vRes, n, chk);
                unsigned long long h[3]; CHECK(hipMemcpyAsync(h, chk, 24, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
                ok = h[0] == 0 && h[1] == ref[1] && h[2] == ref[2];
            }
        }
        allOk = allOk && ok;
        char row[512];
        snprintf(row, sizeof row, "%s{\"mode\": \"%s\", \"variant\": %d, \"ms_best\": %.3f, \"ms_avg\": %.3f, \"ms_per_pass\": %.3f, \"GBps\": %.0f, \"sorted_stable_permutation\": %s}",
                 first ? "" : ", ", m.name, m.variant, best, sumMs / 3, best / 5, 5.0 * n * 24.0 / (best * 1e6), m.sorts ? (ok ? "true" : "false") : "null");
        json += row; first = false;
        fprintf(stderr, "%-28s %8.3f ms best, %8.3f avg, %s\n", m.name, best, sumMs / 3, m.sorts ? (ok ? "ok" : "WRONG") : "-");
    }
    json += "]}";
    printf("%s\n", json.c_str());
    if (outPath) { FILE *f = fopen(outPath, "w"); if (f) { fprintf(f, "%s\n", json.c_str()); fclose(f); } }
    return allOk ? 0 : 2;
}

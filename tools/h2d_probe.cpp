// h2d_probe.cpp -- host-to-device copies of 1.5 GB out of different kinds of host memory (what a batch's bases are held in):
// malloc, an anonymous mapping with and without huge pages, the same registered (hipHostRegister), hipHostMalloc.
//   hipcc -O2 -o tools/h2d_probe tools/h2d_probe.cpp && ./tools/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char *p, size_t n) { std::vector<std::thread> t; for (int i = 0; i < 8; ++i) t.emplace_back([=] { memset(p + n / 8 * i, i + 1, n / 8); }); for (auto &x : t) x.join(); }
int main()
{
    const size_t N = (size_t)1536 << 20;
    char *dev; if (hipMalloc(&dev, N) != hipSuccess) return 1;
    hipStream_t st; hipStreamCreate(&st);
    auto copy = [&](const char *what, char *src) {
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            hipMemcpyAsync(dev, src, N, hipMemcpyHostToDevice, st);
            const double t1 = now();
            hipStreamSynchronize(st);
            const double t2 = now();
            printf("%-40s call %.3f s, sync %.3f s: %.1f GB/s\n", what, t1 - t0, t2 - t1, N / (t2 - t0) / 1e9);
        }
    };
    { char *p = (char *)malloc(N); touch(p, N); copy("malloc", p); free(p); }
    { char *p = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); touch(p, N); copy("mmap, 4K pages", p); munmap(p, N); }
    { char *p = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(p, N, MADV_HUGEPAGE); touch(p, N); copy("mmap, huge pages", p);
      double t0 = now(); hipError_t e = hipHostRegister(p, N, hipHostRegisterDefault); printf("hipHostRegister(huge pages): %s, %.3f s\n", hipGetErrorString(e), now() - t0);
      if (e == hipSuccess) { copy("mmap, huge pages, registered", p); t0 = now(); hipHostUnregister(p); printf("unregister %.3f s\n", now() - t0); }
      munmap(p, N); }
    { char *p = (char *)mmap(nullptr, 2 * N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(p, 2 * N, MADV_HUGEPAGE); touch(p, N); copy("mmap 2x, huge pages, half touched", p); munmap(p, 2 * N); }
    { char *p; double t0 = now(); hipHostMalloc((void **)&p, N, hipHostMallocDefault); printf("hipHostMalloc %.3f s\n", now() - t0); touch(p, N); copy("hipHostMalloc", p); hipHostFree(p); }
    { char *p = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); touch(p, N);
      double t0 = now(); hipError_t e = hipHostRegister(p, N, hipHostRegisterDefault); printf("hipHostRegister(4K pages): %s, %.3f s\n", hipGetErrorString(e), now() - t0);
      if (e == hipSuccess) { copy("mmap, 4K pages, registered", p); hipHostUnregister(p); } munmap(p, N); }
    // one writer of new tmpfs pages: out of page-locked and out of ordinary memory, in calls of different sizes
    {
        const size_t W = (size_t)2 << 30;
        char *pin; hipHostMalloc((void **)&pin, W, hipHostMallocDefault); touch(pin, W);
        char *ord = (char *)mmap(nullptr, W, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(ord, W, MADV_HUGEPAGE); touch(ord, W);
        for (int kind = 0; kind < 2; ++kind)
            for (size_t call : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
                unlink("/dev/shm/h2d_probe.out");
                const int fd = open("/dev/shm/h2d_probe.out", O_WRONLY | O_CREAT | O_TRUNC, 0644);
                const char *src = kind ? ord : pin;
                const double t0 = now();
                for (size_t a = 0; a < W;) { const ssize_t w = pwrite(fd, src + a, std::min(call, W - a), (off_t)a); if (w <= 0) break; a += (size_t)w; }
                printf("one writer, new tmpfs pages, from %s memory, %4zu MB per call: %.2f GB/s\n", kind ? "ordinary   " : "page-locked", call >> 20, W / (now() - t0) / 1e9);
                close(fd);
            }
        unlink("/dev/shm/h2d_probe.out");
    }
    return 0;
}

// write_probe.cpp -- what fills a NEW tmpfs file fastest on this box: one writer, parallel pwrite of fresh pages, fallocate + parallel pwrite, ftruncate + mmap + parallel memcpy, fallocate per chunk (the file-to-file bound of kasa_identify: DESIGN 13 row 7b).  g++ -O2 -pthread -o tools/write_probe tools/write_probe.cpp && tools/write_probe /dev/shm
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void par(unsigned nt, F f) { std::vector<std::thread> pool; for (unsigned t = 0; t < nt; ++t) pool.emplace_back(f, t); for (auto &th : pool) th.join(); }
int main(int argc, char **argv)
{
    const std::string path = std::string(argc > 1 ? argv[1] : "/dev/shm") + "/wprobe.bin";
    const size_t N = (size_t)4 << 30, CH = 64u << 20;
    char *src = (char *)malloc(CH); memset(src, 'x', CH);
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
        unlink(path.c_str());
        int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now();
        par(nt, [&](unsigned t) { for (size_t o = (size_t)t * CH; o < N; o += (size_t)nt * CH) pwrite(fd, src, CH, o); });
        printf("pwrite fresh pages, %u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        close(fd);
    }
    for (unsigned nt : {1u, 4u, 8u, 16u}) {
        unlink(path.c_str());
        int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now();
        posix_fallocate(fd, 0, N);
        double t1 = now();
        par(nt, [&](unsigned t) { for (size_t o = (size_t)t * CH; o < N; o += (size_t)nt * CH) pwrite(fd, src, CH, o); });
        double t2 = now();
        printf("fallocate %.2f GB/s, then pwrite %u threads %.2f GB/s: together %.2f GB/s\n", N / (t1 - t0) / 1e9, nt, N / (t2 - t1) / 1e9, N / (t2 - t0) / 1e9);
        close(fd);
    }
    for (unsigned nt : {1u, 4u, 8u, 16u}) {
        unlink(path.c_str());
        int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now();
        ftruncate(fd, N);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        par(nt, [&](unsigned t) { for (size_t o = (size_t)t * CH; o < N; o += (size_t)nt * CH) memcpy(m + o, src, CH); });
        printf("ftruncate + mmap + memcpy, %u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        munmap(m, N); close(fd);
    }
    for (unsigned nt : {4u, 8u, 16u}) {   // fallocate by several threads on disjoint ranges
        unlink(path.c_str());
        int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now();
        par(nt, [&](unsigned t) { for (size_t o = (size_t)t * CH; o < N; o += (size_t)nt * CH) { fallocate(fd, 0, o, CH); pwrite(fd, src, CH, o); } });
        printf("fallocate + pwrite per chunk, %u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        close(fd);
    }
    for (unsigned nt : {1u, 4u, 8u, 16u}) {   // pages allocated by fallocate, then mapped and filled by several threads (minor faults only)
        unlink(path.c_str());
        int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now();
        posix_fallocate(fd, 0, N);
        double t1 = now();
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        par(nt, [&](unsigned t) { for (size_t o = (size_t)t * CH; o < N; o += (size_t)nt * CH) memcpy(m + o, src, CH); });
        double t2 = now();
        printf("fallocate %.2f GB/s, then mmap + memcpy %u threads %.2f GB/s: together %.2f GB/s\n", N / (t1 - t0) / 1e9, nt, N / (t2 - t1) / 1e9, N / (t2 - t0) / 1e9);
        munmap(m, N); close(fd);
    }
    for (unsigned nt : {4u, 16u}) {   // ... the file allocated piece by piece ahead of the copiers (what a pipeline would do: one thread allocates, others fill)
        unlink(path.c_str());
        int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        double t0 = now();
        ftruncate(fd, N);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<int> ready(N / CH, 0);
        volatile int *rd = ready.data();
        std::thread alloc([&] { for (size_t c = 0; c < N / CH; ++c) { fallocate(fd, 0, c * CH, CH); __atomic_store_n(&rd[c], 1, __ATOMIC_RELEASE); } });
        par(nt, [&](unsigned t) { for (size_t c = t; c < N / CH; c += nt) { while (!__atomic_load_n(&rd[c], __ATOMIC_ACQUIRE)) std::this_thread::yield(); memcpy(m + c * CH, src, CH); } });
        alloc.join();
        printf("one thread fallocates 64 MB pieces ahead, %u threads mmap + memcpy behind it: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        munmap(m, N); close(fd);
    }
    unlink(path.c_str());
    return 0;
}

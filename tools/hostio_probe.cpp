// hostio_probe.cpp -- what the host side of a file-to-file run can move on this box (tmpfs in, tmpfs out, a CPU quota):
// first touch of fresh memory with and without huge pages, pread from the page cache into a buffer that is reused,
// pwrite of new file pages, by 1..16 threads.   g++ -O2 -pthread -o hostio_probe tools/hostio_probe.cpp && ./hostio_probe /dev/shm
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void par(unsigned nt, F f)
{
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) pool.emplace_back(f, t);
    for (auto &th : pool) th.join();
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t GB = (size_t)1 << 30, N = 2 * GB;
    // 1. first touch
    for (int huge = 0; huge < 2; ++huge)
        for (unsigned nt : {1u, 4u, 16u}) {
            char *p = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) madvise(p, N, MADV_HUGEPAGE);
            const double t0 = now();
            par(nt, [&](unsigned t) { const size_t a = N / nt * t; memset(p + a, 1, N / nt); });
            const double t1 = now();
            par(nt, [&](unsigned t) { const size_t a = N / nt * t; memset(p + a, 2, N / nt); });
            const double t2 = now();
            printf("first touch %s, %2u threads: %.2f GB/s; second pass %.2f GB/s\n", huge ? "huge pages" : "4K pages  ", nt, N / (t1 - t0) / 1e9, N / (t2 - t1) / 1e9);
            munmap(p, N);
        }
    // 2. a file in the page cache
    const std::string in = dir + "/hostio_probe.in", out = dir + "/hostio_probe.out";
    char *src = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(src, N, MADV_HUGEPAGE);
    par(16, [&](unsigned t) { memset(src + N / 16 * t, 'A' + (int)t, N / 16); });
    {
        const int fd = open(in.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        for (size_t a = 0; a < N;) { const ssize_t w = write(fd, src + a, std::min<size_t>(N - a, 1 << 30)); if (w <= 0) return 1; a += (size_t)w; }
        close(fd);
    }
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
        const int fd = open(in.c_str(), O_RDONLY);
        const double t0 = now();
        par(nt, [&](unsigned t) { size_t a = N / nt * t, e = a + N / nt; while (a < e) { const ssize_t r = pread(fd, src + a, e - a, (off_t)a); if (r <= 0) break; a += (size_t)r; } });
        printf("pread tmpfs -> warm buffer, %2u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        close(fd);
    }
    {
        const int fd = open(in.c_str(), O_RDONLY);
        for (unsigned nt : {1u, 4u, 16u}) {
            const double t0 = now();
            char *m = (char *)mmap(nullptr, N, PROT_READ, MAP_SHARED, fd, 0);
            std::vector<size_t> cnt(nt);
            par(nt, [&](unsigned t) { size_t c = 0; const char *a = m + N / nt * t, *e = a + N / nt; while (a < e) { const char *q = (const char *)memchr(a, 'Z', (size_t)(e - a)); if (!q) break; ++c; a = q + 1; } cnt[t] = c; });
            printf("mmap tmpfs + memchr over it, %2u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
            munmap(m, N);
        }
        close(fd);
    }
    // 3. new file pages
    for (unsigned nt : {1u, 2u, 4u, 8u, 16u}) {
        unlink(out.c_str());
        const int fd = open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        const double t0 = now();
        par(nt, [&](unsigned t) { size_t a = N / nt * t, e = a + N / nt; while (a < e) { const ssize_t w = pwrite(fd, src + a, std::min<size_t>(e - a, 8u << 20), (off_t)a); if (w <= 0) break; a += (size_t)w; } });
        printf("pwrite new tmpfs pages, %2u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        close(fd);
    }
    for (unsigned nt : {1u, 4u, 16u}) {                       // the same through a mapping of the file (ftruncate + mmap + memcpy)
        unlink(out.c_str());
        const int fd = open(out.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        const double t0 = now();
        if (ftruncate(fd, (off_t)N)) return 1;
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        par(nt, [&](unsigned t) { memcpy(m + N / nt * t, src + N / nt * t, N / nt); });
        printf("mmap new tmpfs file + memcpy, %2u threads: %.2f GB/s\n", nt, N / (now() - t0) / 1e9);
        munmap(m, N);
        close(fd);
    }
    {                                                         // overwrite of pages that exist
        const int fd = open(out.c_str(), O_WRONLY);
        const double t0 = now();
        par(8, [&](unsigned t) { size_t a = N / 8 * t, e = a + N / 8; while (a < e) { const ssize_t w = pwrite(fd, src + a, std::min<size_t>(e - a, 8u << 20), (off_t)a); if (w <= 0) break; a += (size_t)w; } });
        printf("pwrite over existing tmpfs pages, 8 threads: %.2f GB/s\n", N / (now() - t0) / 1e9);
        close(fd);
    }
    unlink(in.c_str()); unlink(out.c_str());
    return 0;
}

/* batch_probe.c -- LD_PRELOAD shim used ONLY by tests/golden/make_fixtures.py to observe where the reference
 * binary cuts its batches.  Compare::CompareWithLib_partialSort zeroes one float row per read of the batch
 * (Utilities::Non_contiguousArray::generate, source/utils/Utilities.hpp:600-618: memset(row, 0, 4 * nTaxa)) and then
 * starts threads for the batch; so counting memset calls of exactly KASA_PROBE_BYTES bytes between thread creations
 * gives the number of reads in every batch.  Nothing of the reference is copied or changed; the shim only counts.
 *
 * With KASA_PROBE_KEEP set, remove() of a file whose name holds that string does nothing: the reference's list of the
 * pieces it reads its input in (Read.hpp:372-600, deleted at Compare.hpp:3694) stays for make_fixtures.py to store.
 *
 *   gcc -O2 -shared -fPIC -o batch_probe.so batch_probe.c -ldl
 *   KASA_PROBE_BYTES=<4*nTaxa> KASA_PROBE_OUT=<file> LD_PRELOAD=./batch_probe.so kASA identify ...
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void *(*real_memset)(void *, int, size_t);
static int (*real_create)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
static size_t probe_bytes;
static unsigned long count, last_logged;
static FILE *out;
static int busy;

static void init(void)
{
    if (busy) return;
    busy = 1;
    real_memset = dlsym(RTLD_NEXT, "memset");
    real_create = dlsym(RTLD_NEXT, "pthread_create");
    const char *b = getenv("KASA_PROBE_BYTES"), *o = getenv("KASA_PROBE_OUT");
    probe_bytes = b ? strtoul(b, 0, 10) : 0;
    if (o) out = fopen(o, "w");
    busy = 0;
}

void *memset(void *p, int c, size_t n)
{
    if (!real_memset) {
        if (busy) { unsigned char *q = p; for (size_t i = 0; i < n; ++i) q[i] = (unsigned char)c; return p; }
        init();
    }
    if (n == probe_bytes && c == 0 && probe_bytes) __sync_fetch_and_add(&count, 1);
    return real_memset(p, c, n);
}

int pthread_create(pthread_t *t, const pthread_attr_t *a, void *(*fn)(void *), void *arg)
{
    if (!real_create) init();
    if (out && count != last_logged) { fprintf(out, "%lu\n", count - last_logged); fflush(out); last_logged = count; }
    return real_create(t, a, fn, arg);
}

int remove(const char *path)
{
    static int (*real_remove)(const char *);
    if (!real_remove) real_remove = dlsym(RTLD_NEXT, "remove");
    const char *keep = getenv("KASA_PROBE_KEEP");
    if (keep && *keep && strstr(path, keep)) return 0;
    return real_remove(path);
}

int unlink(const char *path)
{
    static int (*real_unlink)(const char *);
    if (!real_unlink) real_unlink = dlsym(RTLD_NEXT, "unlink");
    const char *keep = getenv("KASA_PROBE_KEEP");
    if (keep && *keep && strstr(path, keep)) return 0;
    return real_unlink(path);
}

__attribute__((destructor)) static void fini(void)
{
    if (out) { if (count != last_logged) fprintf(out, "%lu\n", count - last_logged); fclose(out); }
}

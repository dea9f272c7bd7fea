/* tools/hsa_passthrough.c -- lets a process with the HOST half of libkasa_hip.so under AddressSanitizer use a GPU.
 *
 * ROCm's clang ships an ASan runtime that intercepts the HSA allocation and copy entry points (its device-side sanitizer
 * wraps device allocations in red zones, which needs xnack+ code objects -- not available on this pool).  A host-only
 * instrumented process never gets that far: the first hipMalloc ends in the interceptor with "allocator is trying to
 * allocate 0x400000 bytes".  This library is preloaded BEFORE the sanitizer runtime and forwards those entry points straight
 * to libhsa-runtime64.so.1, so the interceptors are never reached; everything else of ASan (malloc/free, memcpy and friends,
 * the instrumented host code) works as usual.  Debugging tool only (tools/asan_run.sh).
 *   gcc -O1 -fPIC -shared -I/opt/rocm/include -o tools/libhsa_passthrough.so tools/hsa_passthrough.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

static void *real(const char *name)
{
    static void *lib;
    if (!lib) lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
    void *f = lib ? dlsym(lib, name) : NULL;
    if (!f) { fprintf(stderr, "hsa_passthrough: %s not found in libhsa-runtime64.so.1\n", name); abort(); }
    return f;
}
#define FWD(ret, name, params, args) \
    ret name params { static ret (*f) params; if (!f) f = (ret (*) params)real(#name); return f args; }

FWD(hsa_status_t, hsa_amd_memory_pool_allocate, (hsa_amd_memory_pool_t pool, size_t size, uint32_t flags, void **ptr), (pool, size, flags, ptr))
FWD(hsa_status_t, hsa_amd_memory_pool_free, (void *ptr), (ptr))
FWD(hsa_status_t, hsa_amd_agents_allow_access, (uint32_t n, const hsa_agent_t *agents, const uint32_t *flags, const void *ptr), (n, agents, flags, ptr))
FWD(hsa_status_t, hsa_memory_copy, (void *dst, const void *src, size_t size), (dst, src, size))
FWD(hsa_status_t, hsa_amd_memory_async_copy, (void *dst, hsa_agent_t da, const void *src, hsa_agent_t sa, size_t size, uint32_t nd, const hsa_signal_t *deps, hsa_signal_t done),
    (dst, da, src, sa, size, nd, deps, done))
FWD(hsa_status_t, hsa_amd_memory_async_copy_on_engine, (void *dst, hsa_agent_t da, const void *src, hsa_agent_t sa, size_t size, uint32_t nd, const hsa_signal_t *deps, hsa_signal_t done,
     hsa_amd_sdma_engine_id_t engine, bool force), (dst, da, src, sa, size, nd, deps, done, engine, force))
FWD(hsa_status_t, hsa_amd_ipc_memory_create, (void *ptr, size_t len, hsa_amd_ipc_memory_t *handle), (ptr, len, handle))
FWD(hsa_status_t, hsa_amd_ipc_memory_attach, (const hsa_amd_ipc_memory_t *handle, size_t len, uint32_t n, const hsa_agent_t *agents, void **mapped), (handle, len, n, agents, mapped))
FWD(hsa_status_t, hsa_amd_ipc_memory_detach, (void *mapped), (mapped))
FWD(hsa_status_t, hsa_amd_vmem_address_reserve_align, (void **va, size_t size, uint64_t address, uint64_t alignment, uint64_t flags), (va, size, address, alignment, flags))
FWD(hsa_status_t, hsa_amd_vmem_address_free, (void *va, size_t size), (va, size))

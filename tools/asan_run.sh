#!/bin/bash
# Run a command with the HOST half of the library under AddressSanitizer + UBSan (`make asan` first; device code is the
# product's).  Python is not instrumented itself: the sanitizer runtime is preloaded, every malloc/free of the process goes
# through it, and the instrumented library's own accesses are checked.   tools/asan_run.sh python tools/fuzz_gpu.py 905000 400
R="$(cd "$(dirname "$0")/.." && pwd)"
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
export KASA_LIB="$R/kasa_amd/libkasa_hip_asan.so" KASA_IDENTIFY="$R/kasa_amd/host/kasa_identify_asan"
# detect_leaks=0: CPython and the HIP runtime keep memory until exit; protect_shadow_gap=0: the HSA runtime maps fixed address ranges
# (verify_asan_link_order=0: tools/libhsa_passthrough.so is preloaded before the sanitizer runtime -- see tools/hsa_passthrough.c)
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:print_stacktrace=1:verify_asan_link_order=0:${ASAN_OPTIONS_EXTRA:-}"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0"
[ -f "$R/tools/libhsa_passthrough.so" ] || gcc -O1 -fPIC -shared -I/opt/rocm/include -o "$R/tools/libhsa_passthrough.so" "$R/tools/hsa_passthrough.c" -ldl
LD_PRELOAD="$R/tools/libhsa_passthrough.so:$RT" exec "$@"

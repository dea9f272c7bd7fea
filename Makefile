# Convenience targets; `python -m kasa_amd.build` and `__graft_entry__.build()` do the same compiles.
HIPCC ?= /opt/rocm/bin/hipcc
LIB    = kasa_amd/libkasa_hip.so
HOST   = kasa_amd/host/kasa_identify

all: $(LIB) $(HOST) oracle

$(LIB): kasa_amd/csrc/kasa_hip.hip kasa_amd/csrc/kasa_refbatch.cpp kasa_amd/csrc/stdsort_order.h kasa_amd/csrc/kasa_radix.h kasa_amd/csrc/kasa_text.h kasa_amd/csrc/kasa_replay.h kasa_amd/host/grisu_powers.inc include/kasa_hip.h
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-result -o $@ kasa_amd/csrc/kasa_hip.hip kasa_amd/csrc/kasa_refbatch.cpp -ldl -Wl,-rpath,/opt/rocm/lib

$(HOST): kasa_amd/host/kasa_identify.cpp kasa_amd/host/grisu_powers.inc include/kasa_hip.h $(LIB)
	g++ -O2 -std=c++17 -pthread -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o $@ $< -Lkasa_amd -lkasa_hip -lz -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

# host code under AddressSanitizer + UBSan (device code objects unchanged: -fno-gpu-sanitize); run with tools/asan_run.sh
ASAN_RT = $(shell /opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
asan: kasa_amd/libkasa_hip_asan.so kasa_amd/host/kasa_identify_asan
kasa_amd/libkasa_hip_asan.so: kasa_amd/csrc/kasa_hip.hip kasa_amd/csrc/kasa_refbatch.cpp kasa_amd/csrc/stdsort_order.h kasa_amd/csrc/kasa_radix.h kasa_amd/csrc/kasa_text.h kasa_amd/csrc/kasa_replay.h include/kasa_hip.h
	$(HIPCC) --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer -Wno-unused-result -o $@ kasa_amd/csrc/kasa_hip.hip kasa_amd/csrc/kasa_refbatch.cpp -ldl -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$(dir $(ASAN_RT))
kasa_amd/host/kasa_identify_asan: kasa_amd/host/kasa_identify.cpp kasa_amd/host/grisu_powers.inc include/kasa_hip.h kasa_amd/libkasa_hip_asan.so
	/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -shared-libsan -fno-omit-frame-pointer -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o $@ $< kasa_amd/libkasa_hip_asan.so -lz -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$(dir $(ASAN_RT))

test:
	python -m pytest tests -q -m "not gpu"

test-gpu:
	python -m pytest tests -q -m gpu

clean:
	rm -f $(LIB) $(HOST) kasa_amd/libkasa_hip_asan.so kasa_amd/host/kasa_identify_asan
	$(MAKE) -C oracle clean

.PHONY: all oracle test test-gpu clean asan

"""Build the HIP extension in-tree (kasa_amd/libkasa_hip.so) for gfx950 with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "kasa_hip.hip")
HOST_SRCS = [os.path.join(HERE, "csrc", "kasa_refbatch.cpp")]   # host-only parts of the C ABI
SO = os.path.join(HERE, "libkasa_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "kasa_hip.h")
CSRC_HEADERS = [os.path.join(HERE, "csrc", "stdsort_order.h"), os.path.join(HERE, "csrc", "kasa_radix.h"), os.path.join(HERE, "csrc", "kasa_text.h"), os.path.join(HERE, "csrc", "kasa_replay.h"),
                os.path.join(HERE, "host", "grisu_powers.inc")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def build(force: bool = False) -> str:
    newest = max(os.path.getmtime(p) for p in [SRC, HEADER] + HOST_SRCS + CSRC_HEADERS)
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest:
        return SO
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-result", "-o", SO, SRC] + HOST_SRCS + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return SO


HOST_SRC = os.path.join(HERE, "host", "kasa_identify.cpp")
HOST_BIN = os.path.join(HERE, "host", "kasa_identify")


def build_host(force: bool = False) -> str:
    """The C++ host driver (kASA's `identify` CLI over the C ABI).  KASA_IDENTIFY: another build of it (tools/asan_run.sh)."""
    other = os.environ.get("KASA_IDENTIFY")
    if other and os.path.exists(other):
        return other
    build()
    newest = max(os.path.getmtime(p) for p in (HOST_SRC, HEADER, os.path.join(HERE, "host", "grisu_powers.inc")))
    if not force and os.path.exists(HOST_BIN) and os.path.getmtime(HOST_BIN) >= newest:
        return HOST_BIN
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-o", HOST_BIN, HOST_SRC, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + HERE, "-lkasa_hip", "-lz", "-L/opt/rocm/lib", "-lrccl",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return HOST_BIN


if __name__ == "__main__":
    print(build(force=True))
    print(build_host(force=True))

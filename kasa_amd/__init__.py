"""kasa_amd -- MI355X-native implementation of kASA's `identify` hot path.

The device code (kasa_amd/csrc, built into libkasa_hip.so) sits behind the C ABI of include/kasa_hip.h;
this package is the host side: file formats, read parsing, the ctypes binding, the batch driver that
mirrors Compare::CompareWithLib_partialSort, and the per-read / profile writers.
"""
__all__ = ["formats", "reads", "report", "textnum", "capi", "identify"]

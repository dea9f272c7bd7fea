#!/usr/bin/env python3
"""More seeds of the differential tests than the suite carries (device against the oracle over the option space and on
adversarial query sets): python3 tools/fuzz_gpu.py [first seed] [count] [seconds].  Stops at the first failure."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_parity as T

def fuzz_text(seed):
    """device text against the host's writer (the device's ranking underneath) on a random world"""
    import numpy as np
    from kasa_amd import capi, identify
    rng = np.random.default_rng(70000 + seed)
    ix, batch = T.synthetic_world(int(rng.integers(1, 1 << 30)), int(rng.integers(2, 60)), int(rng.integers(800, 5000)), int(rng.integers(1, 500)),
                                  read_len=int(rng.choice([40, 76, 100, 150, 250])))
    fmt = str(rng.choice(["json", "jsonl", "tsv", "kraken"]))
    beasts = int(rng.choice([1, 2, 3, 5, 10, 50]))
    thr = float(rng.choice([0.0, 0.0, 0.01, 0.05, 0.2]))
    frames = int(rng.choice([1, 3, 6]))
    dix = capi.DeviceIndex(ix)
    texts = []
    per_batch = int(rng.integers(1, batch.n + 1)) if seed % 3 == 0 else None
    for device_text in (True, False):
        run = identify.Identify(ix, 0, 12, 7, frames, thr, beasts, fmt, dix=dix)
        run.device_text = device_text
        text, prof, _ = run.run(batch, True, batch_reads=per_batch)
        texts.append((text, prof))
        run.close()
    dix.close()
    assert texts[0] == texts[1], (seed, fmt, beasts, thr, frames)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 480.0
CYCLE = [0, 1, 2, 4, 262144, 262144 | 1, 2048, 1 | 8192, 1 | 16384 | 8388608, 1 | 1073741824, 1073741824, 134217728]   # (1073741824: sorted-event replay; 134217728: 64-byte cells)
if os.environ.get("FUZZ_NO_THIRD"):
    CYCLE = CYCLE[:-1]
import faulthandler
faulthandler.enable()                                   # a crash inside the library names the Python frame it came from
t0 = time.time()
done = 0
for seed in range(first, first + count):
    if time.time() - t0 > budget:
        break
    if done % 50 == 0:
        print("at seed", seed, file=sys.stderr, flush=True)
    if os.environ.get("FUZZ_VERBOSE"):
        print("seed", seed, file=sys.stderr, flush=True)
    try:
        T.test_random_configurations.__wrapped__(seed) if hasattr(T.test_random_configurations, "__wrapped__") else T.test_random_configurations(seed)
        T.test_adversarial_queries_vs_oracle(seed, CYCLE[seed % len(CYCLE)])   # (262144: long lists by whole wavefronts; 2048: no followers; 8192: lane-owned cells; 16384 | 8388608: the pending window in device memory)
        if os.environ.get("FUZZ_TEXT"):
            fuzz_text(seed)
    except Exception:
        print("FAILED at seed", seed)
        traceback.print_exc()
        sys.exit(1)
    done += 1
print("ok:", done, "seeds from", first, "in %.0f s" % (time.time() - t0))

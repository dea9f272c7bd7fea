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


def fuzz_long(seed):
    """long reads (contigs of pieces of the genomes, regions met again, tandem repeats) against the oracle: the sorted-event replay,
    the encoder's long sequences, long staging rows -- over key widths, k ranges (more than eight levels with either key width),
    frames, -e"""
    import numpy as np
    from kasa_amd import capi, reads
    from oracle import oracle
    rng = np.random.default_rng(123000 + seed)
    K = 25 if seed % 3 == 2 else 12
    k_high = int(rng.integers(max(2, K - 6), K + 1))
    k_low = int(rng.integers(1, min(k_high, 9) + 1))
    if seed % 4 == 0:
        k_high, k_low = K, 7
    frames = int(rng.choice([1, 3, 3, 6]))
    unique = bool(rng.integers(0, 4) == 0)
    n_taxa, glen = int(rng.integers(2, 10)), int(rng.integers(8000, 30000))
    ix, base = T.synthetic_world(int(rng.integers(1, 1 << 30)), n_taxa, glen, 30, K=K)
    pool = base.bases
    seqs = []
    for r in range(int(rng.integers(3, 9))):
        if rng.integers(0, 2) == 0:
            a = int(rng.integers(0, pool.shape[0] - 200)); seqs.append(pool[a:a + int(rng.integers(0, 200))].copy())
            continue
        parts = []
        total, want = 0, int(rng.integers(17000, 60000)) * (3 if frames == 1 else 1)
        while total < want:
            L = int(rng.integers(300, 5000))
            a = int(rng.integers(0, max(1, pool.shape[0] - L)))
            piece = pool[a:a + L].copy()
            kind = int(rng.integers(0, 6))
            if kind == 0 and parts:
                piece = parts[int(rng.integers(0, len(parts)))][:L].copy()       # a region once more
            elif kind == 1:
                piece = np.tile(piece[:int(rng.integers(20, 400))], int(rng.integers(2, 30)))   # a tandem repeat
            parts.append(piece); total += piece.shape[0]
        seqs.append(np.concatenate(parts))
    off = np.concatenate(([0], np.cumsum([x.shape[0] for x in seqs]))).astype(np.int64)
    batch = reads.ReadBatch(np.concatenate(seqs), off, None, np.asarray([x.shape[0] + 1 for x in seqs], dtype=np.uint32))
    flags = int(rng.choice([0, 0, 0, 1, 1073741824, 134217728, 536870912]))
    p = oracle.params(k_high, k_low, frames, K=ix.K)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True, unique=unique, seg_read=None, n_reads=batch.n)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, k_high, k_low, frames)
    ctx.debug_flags(flags)
    ctx.run_batch(batch.bases, batch.offsets, True, unique=unique)
    assert ctx.n_kmers == nq
    st = ctx.batch_stats()
    for k in ("replay_reads", "replay_events", "general_reads", "dense_reads"):
        LONG_STATS[k] = LONG_STATS.get(k, 0) + st[k]
    LONG_STATS["batches"] = LONG_STATS.get("batches", 0) + 1
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    T.assert_csr_equal(T.csr_rows(*ctx.scores()), T.helpers.csr_from_dense(res.M))
    ctx.close(); dix.close()


LONG_STATS = {}


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
        if not os.environ.get("FUZZ_ONLY_LONG"):
            T.test_random_configurations.__wrapped__(seed) if hasattr(T.test_random_configurations, "__wrapped__") else T.test_random_configurations(seed)
        if not os.environ.get("FUZZ_ONLY_LONG"):
            T.test_adversarial_queries_vs_oracle(seed, CYCLE[seed % len(CYCLE)])   # (262144: long lists by whole wavefronts; 2048: no followers; 8192: lane-owned cells; 16384 | 8388608: the pending window in device memory)
        if os.environ.get("FUZZ_TEXT"):
            fuzz_text(seed)
        if os.environ.get("FUZZ_LONG"):
            fuzz_long(seed)
    except Exception:
        print("FAILED at seed", seed)
        traceback.print_exc()
        sys.exit(1)
    done += 1
print("ok:", done, "seeds from", first, "in %.0f s" % (time.time() - t0), LONG_STATS if LONG_STATS else "")

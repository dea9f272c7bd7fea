"""The closed form the device implements must equal the faithful sequential
restatement of Compare.hpp:678-1069 bit for bit -- on adversarial inputs too: tiny alphabets (every
prefix shared), many taxa per k-mer, duplicate queries, '^' letters, queries outside the index."""
import numpy as np
import pytest

from kasa_amd import formats
from oracle import oracle


def random_case(seed, n_idx, n_q, n_taxa, letters, k_high, k_low, n_reads):
    rng = np.random.default_rng(seed)
    letters = np.asarray(letters, dtype=np.uint64)

    def kmers(n):
        v = np.zeros(n, dtype=np.uint64)
        for _ in range(12):
            v = (v << np.uint64(5)) | letters[rng.integers(0, letters.shape[0], size=n)]
        return v
    content = formats.Content(["non_unique"] + [f"T{i}" for i in range(1, n_taxa)],
                              np.arange(n_taxa, dtype=np.uint32) * np.uint32(7))
    ik = kmers(n_idx)
    it = content.taxids[rng.integers(1, n_taxa, size=n_idx)]
    ix = formats.make_index(ik, it, content)
    # queries: half drawn from the index (with small perturbations of the tail), half random
    qa = ix.kmer[rng.integers(0, ix.n, size=n_q // 2)].copy()
    cut = rng.integers(0, 7, size=qa.shape[0]).astype(np.uint64) * np.uint64(5)
    tail = kmers(qa.shape[0])
    qa = ((qa >> cut) << cut) | (tail & ((np.uint64(1) << cut) - np.uint64(1)))
    q = np.concatenate((qa, kmers(n_q - qa.shape[0])))
    q = np.concatenate((q, q[: n_q // 5]))  # exact duplicates
    rd = rng.integers(0, n_reads, size=q.shape[0]).astype(np.uint32)
    return ix, q, rd


@pytest.mark.parametrize("seed", range(40))
def test_closed_form_equals_sequential(seed):
    rng = np.random.default_rng(1000 + seed)
    letters = [[1, 2], [1, 2, 30], [3, 4, 5, 30, 31], list(range(1, 21))][seed % 4]
    k_low = int(rng.integers(6, 12))
    k_high = int(rng.integers(k_low, 13))
    n_taxa = int(rng.integers(2, 12))
    n_reads = int(rng.integers(1, 9))
    ix, q, rd = random_case(seed, int(rng.integers(1, 400)), int(rng.integers(5, 600)), n_taxa, letters,
                            k_high, k_low, n_reads)
    p = oracle.params(k_high, k_low, 3)
    iv = oracle.IndexView(ix)
    q, rd = oracle.sort_queries(q, rd)
    rs, rl = oracle.ranges(iv, p, q)
    a = oracle.compare(iv, p, q, rd, rs, rl, n_reads, True, closed_form=False)
    b = oracle.compare(iv, p, q, rd, rs, rl, n_reads, True, closed_form=True)
    assert np.array_equal(a.count_unique, b.count_unique)
    assert np.array_equal(a.count_all.view(np.uint64), b.count_all.view(np.uint64))
    assert np.array_equal(a.M.view(np.uint32), b.M.view(np.uint32))
    # profile-only mode gives the same tables
    c = oracle.compare(iv, p, q, rd, rs, rl, n_reads, False, closed_form=False)
    assert np.array_equal(a.count_unique, c.count_unique)
    assert np.array_equal(a.count_all.view(np.uint64), c.count_all.view(np.uint64))


def test_threaded_batch_equals_the_single_threaded_one():
    """ko_identify_threaded (bench.py's cpu_baseline: the reference's -n threading model) computes the same profile as
    the plain sequence encode -> sort -> ranges -> compare; per-read floats are racy there as in the reference, so only
    the order-independent tables are compared."""
    import os
    from kasa_amd import reads
    from tests import helpers
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    p = oracle.params(12, 7, 3)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    iv = oracle.IndexView(ix)
    for threads in (1, 3, 8):
        ca, cu, n = oracle.identify_threaded(iv, batch.bases, batch.offsets, p, threads, want_tables=True)
        assert n == nq
        assert np.array_equal(cu, res.count_unique)
        np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)

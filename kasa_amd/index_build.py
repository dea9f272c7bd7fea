"""`kASA build` for the device path: a reference FASTA + content file -> the index
files `identify` reads, produced with the device's own encoder and radix sort.

What the reference's build mode leaves on disk (source/modes/Build.hpp:305-477, Trie.hpp:365-394,
kASA.hpp:449-575) is, per database sequence, every 3-frame window of K codons *including* the windows that run
over the end of the sequence (padded with '^' letters, down to a single real letter), tagged with the
sequence's tax ID, sorted by (k-mer, tax ID) and made unique.  That is exactly what the read encoder emits for
kLow = 1 (marker of 3(K-1) `X` bases), so the build is: encode with kLow = 1 -> sort -> unique.
"""
from __future__ import annotations

import numpy as np

from . import capi, formats, reads


def build_index(fasta_path: str, content_path: str, device: int = 0, K: int = formats.K64, codon_lut=None) -> formats.Index:
    content = formats.read_content(content_path)
    acc_to_tax = {}
    with open(content_path) as f:                                 # column 4: accessions of the taxon, ';'-separated
        for line in f:
            cols = line.rstrip("\n").split("\t")
            if len(cols) >= 4:
                tid = int(cols[4]) if len(cols) >= 5 else int(cols[1])
                for acc in cols[3].split(";"):
                    acc_to_tax[acc] = tid
    db = reads.parse_reads(fasta_path)
    tax_of_seq = np.zeros(db.n, dtype=np.uint32)
    for i, name in enumerate(db.names):
        acc = name.split(" ")[0]
        if acc not in acc_to_tax:
            raise RuntimeError("sequence " + acc + " is not listed in the content file")
        tax_of_seq[i] = acc_to_tax[acc]
    one = np.array([1], dtype=np.uint64)
    if K > formats.K64:
        one = np.zeros(1, dtype=formats.KEY128_DTYPE)
        one["lo"] = 1
    boot = formats.make_index(one, content.taxids[1:2].copy(), content)
    dix = capi.DeviceIndex(boot, device, check_trie=False)
    ctx = capi.Context(dix, K, 1, 3, codon_lut)                   # kLow = 1: every tail window, '^'-padded
    ctx.upload(db.bases, db.offsets)
    ctx.encode()
    ctx.sort_and_range()
    km, seq = ctx.queries()
    ctx.close()
    dix.close()
    return formats.make_index(km, tax_of_seq[seq], content)       # sort by (k-mer, tax ID) + unique + trie + frequencies

"""Host side of ``mustache recall``: same function name, arguments and TSV as the reference's ``_recall``
(reference mustache/recall.py:15-28).  For every position named in a pairflanks table the reads around it are tallied
again (clipped 5p / 3p, run-through, small / large insertions, deletions, deletions one base up / down) -- the
per-site read join of findflanks at GIVEN sites, without consensus.  The tallies run in libmfk.so (``mfk_recall``).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import lib as _lib
from .findflanks import device_reads, log
from .pack import pack_bam

RECALL_COLUMNS = ["contig", "pos", "orient", "softclip_count_5p", "softclip_count_3p", "runthrough_count",
                  "small_insertion_count_5p", "small_insertion_count_3p", "large_insertion_count_5p",
                  "large_insertion_count_3p", "deletion_count", "upstream_deletion_count", "downstream_deletion_count",
                  "total_count"]   # make_dataframe, recall.py:117-120


def recall_sites(pairs):
    """Sites in the reference's order: contigs as they first appear in the pairs table (load_pairs fills a dict row by
    row, recall.py:44-47), positions ascending inside a contig (make_dataframe, recall.py:123-126), each once."""
    by_contig = {}
    for contig, p5, p3 in zip(pairs["contig"], pairs["pos_5p"], pairs["pos_3p"]):
        d = by_contig.setdefault(contig, set())
        d.add(int(p5))
        d.add(int(p3))
    return [(c, p) for c, ps in by_contig.items() for p in sorted(ps)]


def recall_frame(sites, counts):
    rows = []
    for (contig, pos), c in zip(sites, counts):
        c = [int(x) for x in c]
        total = sum(c[:8])                                    # SoftclipSite.get_total_count, findflanks.py:789-795
        for orient in ("5p", "3p"):                           # recall.py:129-143: the same numbers under both orients
            rows.append([contig, pos, orient] + c + [total])
    return pd.DataFrame(rows, columns=RECALL_COLUMNS)


def recall_packed(binding, reads_struct, contig_names, params, pairs, output_file=None, verbose=True):
    """The part of ``_recall`` after the BAM has been packed and moved to the device."""
    sites = recall_sites(pairs)
    ids = {str(n): i for i, n in enumerate(contig_names)}
    for contig, _ in sites:
        if str(contig) not in ids:
            raise KeyError("contig %r of the pairs table is not in the BAM header" % (contig,))
    if verbose:
        log("Getting clipped and unclipped read information near softclipped sites...")
    counts = binding.run_recall(reads_struct, params, [ids[str(c)] for c, _ in sites], [p for _, p in sites])
    out = recall_frame(sites, counts)
    if output_file:
        if verbose:
            log("Saving results to file %s" % output_file)
        out.to_csv(output_file, sep="\t", index=False)      # recall.py:26
    return out


def _recall(pairsfile, bamfile, min_alignment_quality, min_alignment_inner_length, large_insertion_cutoff, output_file,
            device="cuda:0", verbose=True, decoder="gpu"):
    """Drop-in for reference recall.py:15-28 (same positional arguments).  ``decoder`` as in ``_findflanks``: "gpu" (the BAM is
    decoded on the device, gpudecode.py) or "native" (host threads, libmfkio.so); same arrays."""
    pairs = pd.read_csv(pairsfile, sep="\t")
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    params = _lib.make_params(min_alignment_quality=min_alignment_quality,
                              min_alignment_inner_length=min_alignment_inner_length,
                              large_insertion_cutoff=large_insertion_cutoff)
    packed = None
    if decoder == "gpu":
        from .gpudecode import GpuDecodeUnverified, decode_bam
        try:
            packed = decode_bam(binding, bamfile, pool="none", device=device)
        except GpuDecodeUnverified as e:
            log("device decode: %s; reading the file with the host reader instead" % e)
    elif decoder != "native":
        raise ValueError("decoder must be 'gpu' or 'native', not %r" % (decoder,))
    if packed is None:
        packed = pack_bam(bamfile, pool="none")        # libmfkio.so: multi-threaded BGZF inflate + packing
    rs = device_reads(binding, packed, device, clip_rec=False)
    return recall_packed(binding, rs, packed.contig_names, params, pairs, output_file, verbose)

"""Large synthetic BAM files for the end-to-end tests and the decode benchmark.

A simulated read set (``simulate.simulate_readset``) is written ``copies`` times, copy k on its own set of contigs
(``<name>_<k>``) with its own query names (``<name>#<k>``).  Every copy is an independent genome with identical
alignments, so the findflanks / pairflanks tables of the big file are those of the base set repeated per copy: a
size-independent property a test can check at any size.  Copies are produced by patching the reference ids and the
name suffix in one pre-encoded byte blob (numpy) and compressing in worker processes, so tens of millions of records
take seconds per million instead of the minutes the record-by-record ``BamWriter`` would.
"""
from __future__ import annotations

import multiprocessing as mp
import struct

import numpy as np

from .bamio import _BGZF_EOF, bam_header_bytes, bgzf_block, encode_record

_DIGITS = 6
_BLOCK = 0xFF00
_state = {}


def _encode_base(readset):
    """(blob uint8, positions of the ref id fields, base ref ids, positions of the first name-suffix digit)."""
    chunks, ref_pos, ref_ids, dig_pos, off = [], [], [], [], 0
    for r in readset:
        cid = r.reference_id if r.reference_name is not None else -1
        rec = encode_record(r, cid, name=r.query_name + "#" + "0" * _DIGITS)
        chunks.append(rec)
        ref_pos.append(off + 4)
        ref_ids.append(cid)
        dig_pos.append(off + 36 + len(r.query_name.encode()) + 1)
        off += len(rec)
    blob = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    return blob, np.asarray(ref_pos, np.int64), np.asarray(ref_ids, np.int64), np.asarray(dig_pos, np.int64)


def _copy_bytes(k):
    """BGZF blocks of copy k."""
    blob, ref_pos, ref_ids, dig_pos, n_contigs, level = (_state[x] for x in ("blob", "ref_pos", "ref_ids", "dig_pos",
                                                                            "n_contigs", "level"))
    b = blob.copy()
    ids = np.where(ref_ids >= 0, ref_ids + k * n_contigs, ref_ids).astype("<i4")
    idb = ids.view(np.uint8).reshape(-1, 4)
    for j in range(4):
        b[ref_pos + j] = idb[:, j]
    digits = np.frombuffer(("%0*d" % (_DIGITS, k)).encode(), dtype=np.uint8)
    for j in range(_DIGITS):
        b[dig_pos + j] = digits[j]
    raw = b.tobytes()
    return b"".join(bgzf_block(raw[o:o + _BLOCK], level) for o in range(0, len(raw), _BLOCK))


def replicated_contigs(contigs, copies):
    return [("%s_%0*d" % (n, _DIGITS, k), l) for k in range(copies) for n, l in contigs]


def write_replicated_bam(path, readset, copies, procs=8, level=1):
    """Write ``copies`` copies of ``readset`` to ``path``; returns (contigs of the file, records written)."""
    if copies >= 10 ** _DIGITS:
        raise ValueError("too many copies")
    blob, ref_pos, ref_ids, dig_pos = _encode_base(readset)
    _state.update(blob=blob, ref_pos=ref_pos, ref_ids=ref_ids, dig_pos=dig_pos, n_contigs=len(readset.contigs),
                  level=level)
    contigs = replicated_contigs(readset.contigs, copies)
    head = bam_header_bytes(contigs)
    with open(path, "wb") as fh:
        for o in range(0, len(head), _BLOCK):
            fh.write(bgzf_block(head[o:o + _BLOCK], level))
        if procs > 1 and copies > 1:
            with mp.get_context("fork").Pool(min(procs, copies)) as pool:   # workers inherit _state
                for data in pool.imap(_copy_bytes, range(copies), chunksize=1):
                    fh.write(data)
        else:
            for k in range(copies):
                fh.write(_copy_bytes(k))
        fh.write(_BGZF_EOF)
    _state.clear()
    return contigs, copies * len(ref_pos)


def replicate_table(df, copies, contig_col="contig"):
    """The table a per-copy result implies for the whole file: rows of ``df`` repeated per copy, contigs renamed."""
    import pandas as pd
    parts = []
    for k in range(copies):
        d = df.copy()
        d[contig_col] = ["%s_%0*d" % (c, _DIGITS, k) for c in d[contig_col]]
        parts.append(d)
    return pd.concat(parts, ignore_index=True) if parts else df.copy()


del struct

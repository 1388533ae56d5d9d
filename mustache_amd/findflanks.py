"""Host side of ``mustache findflanks``: same function name, arguments and TSV as the reference's stage driver
``_findflanks`` (reference mustache/findflanks.py:18-59); the per-read and per-site work runs in libmfk.so on the GPU.

Flow: BAM decode + packing on the CPU (``pack_bam``: libmfkio.so, multi-threaded BGZF inflate; SoA, 22 B/read) ->
device tensors -> ``mfk_findflanks`` -> flank rows -> pandas frame with the reference's columns -> ``to_csv(sep='\\t', index=False)``.
"""
from __future__ import annotations

import sys

import numpy as np
import pandas as pd

from . import lib as _lib
from .pack import PackedReads, pack_bam, pack_reads

FLANK_COLUMNS = ["contig", "pos", "orient", "softclip_count_5p", "softclip_count_3p", "runthrough_count",
                 "small_insertion_count_5p", "small_insertion_count_3p", "large_insertion_count_5p",
                 "large_insertion_count_3p", "deletion_count", "upstream_deletion_count", "downstream_deletion_count",
                 "total_count", "consensus_softclip_count", "consensus_seq"]  # make_dataframe, findflanks.py:633-637

_STAT_MESSAGES = [
    (1, "After filtering by minimum softclip length of {min_softclip_length:d}, {n:d} sites remain"),
    (2, "After filtering by minimum softclipped read count of {min_softclip_count:d}, {n:d} sites remain"),
    (3, "After filtering by minimum nearest mate distance {min_distance_to_mate:d}, {n:d} sites remain"),
    (4, "After filtering by minimum softclip ratio of {min_softclip_ratio:f} and a maximum indel ratio of "
        "{max_indel_ratio:f}, {n:d} sites remain"),
    (5, "After filtering by minimum nearest mate distance {min_distance_to_mate:d}, {n:d} sites remain"),
    (6, "After filtering consensus sequences by a minimum length of {min_softclip_count:d}, {n:d} flank sequences "
        "remain"),
    (7, "After filtering consensus sequences by a minimum softclip count of {min_softclip_count:d}, {n:d} flank "
        "sequences remain"),
    (8, "After selecting the best consensus sequence at each site, {n:d} flank sequences remain."),
    (9, "After filtering by minimum nearest mate distance {min_distance_to_mate:d}, {n:d} sites remain"),
]


def log(msg):
    print("[mustache_amd] " + msg, file=sys.stderr)


LAST_TIMING = {}     # stage split of the last _findflanks / _pairflanks call in this process (seconds), for bench.py's end_to_end line


def index_path(table_file, enabled=True):
    """Where the record index of a findflanks run lives: next to its TSV (which is pairflanks' input)."""
    import os
    if not enabled or not table_file or os.environ.get("MFK_NO_RECORD_INDEX", "0") not in ("", "0"):
        return None
    return str(table_file) + ".mfki"


def device_reads(binding, packed: PackedReads, device, clip_rec=True):
    """Move a PackedReads to ``device`` and describe it to the library.  clip_rec: attach the dense clip records
    (``PackedReads.clip_records``, include/mfk.h mfk_clip_rec): worth their one host pass when the reads stay resident for
    more than one call; the one-shot commands, whose time is the decode, leave them out."""
    if hasattr(packed, "device_struct"):        # decoded on the device (gpudecode.DevicePacked): the columns are there already
        return packed.device_struct(binding)
    tensors = packed.to_torch(device, clip_rec=clip_rec)
    return binding.reads_struct(tensors, packed.n_reads, packed.max_span, _lib.host_header(packed))


def flank_frame(result, contig_names):
    """Result of ``Binding.run_findflanks`` -> DataFrame with the reference's columns and row order."""
    fl, seqs = result["flanks"], result["seqs"]
    if len(fl) == 0:
        return pd.DataFrame(columns=FLANK_COLUMNS)
    text = bytes(seqs).decode("ascii")
    data = {"contig": np.asarray(contig_names, dtype=object)[fl["contig_id"]], "pos": fl["pos"].astype(np.int64),
            "orient": np.where(fl["orient"] == 5, "5p", "3p").astype(object)}
    for c in FLANK_COLUMNS[3:15]:
        data[c] = fl[c].astype(np.int64)
    data["consensus_seq"] = [text[o:o + n] for o, n in zip(fl["seq_off"].tolist(), fl["seq_len"].tolist())]
    return pd.DataFrame(data, columns=FLANK_COLUMNS)


def findflanks_packed(binding, reads_struct, contig_names, params, sample_id, output_file=None, verbose=True):
    """The part of ``_findflanks`` after the BAM has been packed and moved to the device."""
    try:
        res = binding.run_findflanks(reads_struct, params)
    except _lib.MfkError as e:
        if "READ NUMBER ERROR" in str(e):
            # the reference's fatal invariant (findflanks.py:971-978, 1025-1027): message on stdout, then sys.exit()
            print("READ NUMBER ERROR - ZERO OR MORE THAN TWO READS ASSOCIATED WITH QUERY NAME")
            sys.exit()
        raise
    if verbose:
        pd_ = {k: getattr(params, k) for k, _ in params._fields_}
        for idx, msg in _STAT_MESSAGES:
            log(msg.format(n=int(res["stats"][idx]), **pd_))
    final_df = flank_frame(res, contig_names)
    final_df.insert(0, "flank_id", list(final_df.index + 1))   # findflanks.py:52
    final_df.insert(0, "sample", sample_id)                     # findflanks.py:53
    if output_file:
        if verbose:
            log("Saving results to file %s" % output_file)
        final_df.to_csv(output_file, sep="\t", index=False)     # findflanks.py:57
    return final_df


def _findflanks(bamfile, min_softclip_length, min_softclip_count, min_alignment_quality, min_alignment_inner_length,
                min_distance_to_mate, min_softclip_ratio, max_indel_ratio, large_insertion_cutoff,
                min_count_consensus, sample_id, output_file=None, device="cuda:0", verbose=True, decoder="gpu",
                exclude_flags=0, record_index=True):
    """Drop-in for reference findflanks.py:18-59 (same positional arguments).  ``decoder``: "gpu" (the default; mfk_decode_bam: inflate, record
    boundaries, packing and name links as kernels, only the compressed file crosses PCIe; a file whose guessed record boundaries do
    not verify is read by the host reader instead), "native" (libmfkio.so, host threads; same arrays) or "python" (bamio.BamReader + pack_reads, the
    readable statement of the same packing rules).  ``exclude_flags``: skip
    records with any of these FLAG bits (0x900 = what the reference's formatbam step removes, samtools.py:9-18), for BAM
    files that did not go through formatbam; 0 = every mapped record, like the reference.  ``record_index``: the pass over
    the BAM also leaves ``<output_file>.mfki`` (record offsets + coordinates per 32 KiB of the uncompressed stream and the
    BGZF block table, include/mfk_bamio.h), through which ``_pairflanks`` reads only the blocks around its candidate
    regions instead of inflating the file a second time -- what the reference's bam.fetch gets from the BAI
    (pairflanks.py:393-409).  MFK_NO_RECORD_INDEX=1 or record_index=False turn it off."""
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    params = _lib.make_params(min_softclip_length=min_softclip_length, min_softclip_count=min_softclip_count,
                              min_alignment_quality=min_alignment_quality,
                              min_alignment_inner_length=min_alignment_inner_length,
                              min_distance_to_mate=min_distance_to_mate, min_softclip_ratio=min_softclip_ratio,
                              max_indel_ratio=max_indel_ratio, large_insertion_cutoff=large_insertion_cutoff,
                              min_count_consensus=min_count_consensus)
    import time
    t0 = time.perf_counter()
    if verbose:
        log("Parsing softclipped sites from provided BAM file...")
    if decoder == "native":
        packed = pack_bam(bamfile, pool="clipped", exclude_flags=exclude_flags, index_out=index_path(output_file, record_index))
    elif decoder == "gpu":
        # the whole decode on the device (gpudecode.py, mfk_decode_bam): only the compressed file crosses PCIe
        from .gpudecode import GpuDecodeUnverified, decode_bam
        try:
            packed = decode_bam(binding, bamfile, pool="clipped", exclude_flags=exclude_flags, device=device, index_out=index_path(output_file, record_index))
        except GpuDecodeUnverified as e:
            log("device decode: %s; reading the file with the host reader instead" % e)
            packed = pack_bam(bamfile, pool="clipped", exclude_flags=exclude_flags, index_out=index_path(output_file, record_index))
    elif decoder == "python":
        from .bamio import BamReader
        with BamReader(bamfile) as bam:
            packed = pack_reads(bam.contigs, bam, pool="clipped", exclude_flags=exclude_flags)
    else:
        raise ValueError("decoder must be 'native', 'gpu' or 'python', not %r" % (decoder,))
    if verbose:
        log("Packed %d mapped reads (%.1f MB of records)" % (packed.n_reads, packed.nbytes_hot() / 1e6))
        st = getattr(packed, "io_stats", None)
        if st and st.get("decoder") == "gpu":
            log("BAM decode on the device: %.3f s (block table + header %.3f, file upload %.3f, inflate + boundaries + count %.3f, fill + links %.3f), %.1f M records/s" % (
                st["seconds_total"], st["seconds_block_table_and_header"], st["seconds_upload_file"], st["seconds_device_stage1"], st["seconds_device_stage2"],
                st["n_records"] / max(st["seconds_total"], 1e-9) / 1e6))
        elif st:
            log("BAM decode: %.2f s on %d threads (inflate %.2f s, parse %.2f s, names %.2f s), %.1f M records/s" % (
                st["seconds_total"], st["n_threads"], st["seconds_inflate"], st["seconds_parse"], st["seconds_link"],
                st["n_records"] / max(st["seconds_total"], 1e-9) / 1e6))
    t1 = time.perf_counter()
    rs = device_reads(binding, packed, device, clip_rec=False)
    if str(device).startswith("cuda"):
        import torch
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = findflanks_packed(binding, rs, packed.contig_names, params, sample_id, output_file, verbose)
    t3 = time.perf_counter()
    st = getattr(packed, "io_stats", None) or {}
    LAST_TIMING["findflanks"] = {"decode_s": round(t1 - t0, 4), "inflate_s": round(st.get("seconds_inflate", 0.0), 4),
                                 "parse_s": round(st.get("seconds_parse", 0.0), 4), "links_s": round(st.get("seconds_link", 0.0), 4),
                                 "upload_s": round(t2 - t1, 4), "kernels_and_tsv_s": round(t3 - t2, 4), "device_ms": round(binding.device_ms(), 3),
                                 "reads_packed": int(packed.n_reads), "inflated_bytes": int(st.get("uncompressed_bytes", 0)), "decoder": st.get("decoder", decoder)}
    for k in ("seconds_block_table_and_header", "seconds_upload_file", "seconds_device_stage1", "seconds_device_stage2"):
        if k in st:
            LAST_TIMING["findflanks"][k.replace("seconds_", "") + "_s"] = round(st[k], 4)
    return out

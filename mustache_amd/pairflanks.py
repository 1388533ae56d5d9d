"""Host side of ``mustache pairflanks``: same function name, arguments and TSV as the reference's ``_pairflanks``
(reference mustache/pairflanks.py:18-51); window join, inverted-repeat test, greedy assignment, spanning-read count,
junction filter and direct-repeat pileup run in libmfk.so on the GPU (``mfk_pairflanks``).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import lib as _lib
from .findflanks import device_reads, index_path, log
from .pack import pack_bam, pack_reads

PAIR_COLUMNS = ["pair_id", "contig", "pos_5p", "pos_3p", "softclip_count_5p", "softclip_count_3p", "total_count_5p",
                "total_count_3p", "spanning_count", "has_IR", "IR_length", "IR_5p", "IR_3p", "seq_5p", "seq_3p",
                "direct_repeat_reference", "direct_repeat_reads_consensus"]  # get_header_list, pairflanks.py:125-130

TARGET_REGION_SIZE = 50  # get_read_direct_repeats default, pairflanks.py:288


def flanks_to_table(flanks: pd.DataFrame, contig_names):
    """Flank frame (as read from the findflanks TSV) -> (FLANK_DTYPE array, seq bytes).  Unknown contigs are an error
    exactly where the reference would fail too (bam.fetch on a contig that is not in the header)."""
    index = {n: i for i, n in enumerate(contig_names)}
    tab = np.zeros(len(flanks), dtype=_lib.FLANK_DTYPE)
    seqs, off = [], 0
    for k, row in enumerate(flanks.itertuples(index=False)):
        contig = str(row.contig)
        if contig not in index:
            raise KeyError("contig %r of the flanks file is not in the BAM header" % contig)
        t = tab[k]
        t["contig_id"], t["pos"] = index[contig], int(row.pos)
        t["orient"] = 5 if row.orient == "5p" else (3 if row.orient == "3p" else 0)
        for c in ("softclip_count_5p", "softclip_count_3p", "runthrough_count", "small_insertion_count_5p",
                  "small_insertion_count_3p", "large_insertion_count_5p", "large_insertion_count_3p", "deletion_count",
                  "upstream_deletion_count", "downstream_deletion_count", "total_count", "consensus_softclip_count"):
            t[c] = int(getattr(row, c))
        s = str(row.consensus_seq).encode("ascii")
        t["seq_len"], t["seq_off"] = len(s), off
        seqs.append(s)
        off += len(s)
    return tab, b"".join(seqs)


def candidate_regions(tab, max_direct_repeat_length, genome_len):
    """Per contig, the reference intervals pairflanks can touch: the spanning fetch [pos3-1, pos5+2) and the direct
    repeat window (pairflanks.py:294-305) of every (5p, 3p) candidate.  Returns (contig_id, start, end) arrays."""
    out = []
    five = tab[tab["orient"] == 5]
    three = tab[tab["orient"] == 3]
    for f5 in five:
        m = three[(three["contig_id"] == f5["contig_id"]) & (three["pos"] >= f5["pos"] - max_direct_repeat_length - 1)
                  & (three["pos"] < f5["pos"])]
        for f3 in m:
            start, end = int(f3["pos"]), int(f5["pos"])
            center = round((end + start) / 2)
            es = max(int(center - TARGET_REGION_SIZE / 2), 0)
            ee = min(int(center + TARGET_REGION_SIZE / 2), genome_len[int(f5["contig_id"])])
            out.append((int(f5["contig_id"]), min(es, start - 1), max(ee, end + 2), es, ee))
    return out


def pair_frame(result, flank_tab, flank_seqs, contig_names):
    pr, strs = result["pairs"], result["strs"]

    def s(off, n):
        return None if n < 0 else strs[int(off):int(off) + int(n)].decode("ascii")
    data = {c: [] for c in PAIR_COLUMNS}
    for k, row in enumerate(pr):
        f5, f3 = flank_tab[int(row["index_5p"])], flank_tab[int(row["index_3p"])]
        data["pair_id"].append(k + 1)                                         # pairflanks.py:120
        data["contig"].append(contig_names[int(row["contig_id"])])
        for c in ("pos_5p", "pos_3p", "softclip_count_5p", "softclip_count_3p", "total_count_5p", "total_count_3p",
                  "spanning_count"):
            data[c].append(int(row[c]))
        data["has_IR"].append(bool(row["has_ir"]))
        data["IR_length"].append(int(row["ir_length"]))
        data["IR_5p"].append(s(row["ir5_off"], row["ir5_len"]))
        data["IR_3p"].append(s(row["ir3_off"], row["ir3_len"]))
        data["seq_5p"].append(flank_seqs[int(f5["seq_off"]):int(f5["seq_off"]) + int(f5["seq_len"])].decode("ascii"))
        data["seq_3p"].append(flank_seqs[int(f3["seq_off"]):int(f3["seq_off"]) + int(f3["seq_len"])].decode("ascii"))
        data["direct_repeat_reference"].append(s(row["dr_ref_off"], row["dr_ref_len"]))
        data["direct_repeat_reads_consensus"].append(s(row["dr_reads_off"], row["dr_len"]))
    df = pd.DataFrame(data, columns=PAIR_COLUMNS)
    if len(df):
        df["has_IR"] = df["has_IR"].astype(bool)
    return df


def genome_arrays(genome, contig_names):
    """Concatenate the FASTA records in BAM contig order -> (uint8 array, int64 offsets).  A contig missing from the
    FASTA gets length 0 (the reference raises KeyError only if a pair lands on it; so do we, see pairflanks_packed)."""
    chunks, off = [], [0]
    for n in contig_names:
        s = genome.get(n, "")
        chunks.append(np.frombuffer(s.encode("ascii"), dtype=np.uint8))
        off.append(off[-1] + len(s))
    g = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    if g.size == 0:
        g = np.zeros(1, np.uint8)
    return g, np.asarray(off, dtype=np.int64)


def contig_ranks(contig_names):
    order = sorted(range(len(contig_names)), key=lambda i: contig_names[i])
    rank = np.zeros(len(contig_names), dtype=np.int32)
    for r, i in enumerate(order):
        rank[i] = r
    return rank


def pairflanks_packed(binding, reads_struct, contig_names, params, flanks: pd.DataFrame, genome_dev, genome_off,
                      output_file=None, verbose=True, ir_backend="builtin", ir_detector=None):
    """``_pairflanks`` after the inputs are on the device.  ``flanks`` is the frame read from the findflanks TSV.
    ``ir_backend``: "builtin" = the library's own inverted-repeat detector, "einverted" = run EMBOSS
    einverted per candidate pair like the reference does and pass its repeats to the library."""
    if flanks.shape[0] == 0:                                                   # pairflanks.py:32-40 (quirk P12)
        if verbose:
            log("No flanks found in the input file...")
        flank_pairs = pd.DataFrame(columns=PAIR_COLUMNS)
        flank_pairs.insert(0, "sample", None)
        if output_file:
            flank_pairs.to_csv(output_file, sep="\t", index=False)
        return flank_pairs
    tab, seqs = flanks_to_table(flanks, contig_names)
    if ir_backend == "einverted":
        from . import einverted
        binding.set_pair_ir(*einverted.ir_table(tab, seqs, int(params.max_direct_repeat_length), detector=ir_detector))
    elif ir_backend != "builtin":
        raise ValueError("ir_backend must be 'builtin' or 'einverted', not %r" % (ir_backend,))
    res = binding.run_pairflanks(reads_struct, params, tab, seqs, contig_ranks(contig_names), genome_dev, genome_off)
    if verbose:
        st = res["stats"]
        log("Finding all inverted repeats at termini in %d candidate pairs..." % st[0])
        log("Identified %d flank pairs in total..." % st[1])
        log("Identified %d flank pairs with inverted repeats..." % st[2])
        log("Identified %d flank pairs with reads that span the insertion..." % st[3])
        log("%d flank pairs remain after filtering..." % st[4])
    flank_pairs = pair_frame(res, tab, seqs, contig_names)
    sample_id = list(flanks["sample"])[0]                                      # pairflanks.py:44
    flank_pairs.insert(0, "sample", sample_id)
    if output_file:
        if verbose:
            log("Saving results to file %s" % output_file)
        flank_pairs.to_csv(output_file, sep="\t", index=False)
    return flank_pairs


def _pack_regions_python(bam, contig_names, regions, exclude_flags=0):
    """The region filter + pool rule of pack_bam(pool='regions') spelled out on ``AlignedRead`` objects."""
    by_contig = {}
    for cid, lo, hi, es, ee in regions:
        by_contig.setdefault(cid, []).append((lo, hi, es, ee))

    def near(read, cid):
        for lo, hi, _, _ in by_contig.get(cid, ()):
            if read.reference_start < hi and read.reference_end > lo:
                return True
        return False

    def in_window(read, cid):
        for _, _, es, ee in by_contig.get(cid, ()):
            if read.reference_start < ee and read.reference_end > es:
                return True
        return False
    index = {n: i for i, n in enumerate(contig_names)}
    reads = (r for r in bam if r.cigartuples is not None and near(r, index[r.reference_name]))
    return pack_reads(bam.contigs, reads, pool=in_window, exclude_flags=exclude_flags)


def _pairflanks(flanksfile, bamfile, genome, max_direct_repeat_length, min_alignment_quality,
                min_alignment_inner_length, max_junction_spanning_prop=0.15, large_insertion_cutoff=30,
                output_file=None, device="cuda:0", verbose=True, ir_backend="builtin", decoder="native", exclude_flags=0,
                record_index=True):
    """Drop-in for reference pairflanks.py:18-51 (same positional arguments).  ``decoder`` as in ``_findflanks``.
    ``record_index``: use ``<flanksfile>.mfki`` when ``_findflanks`` left it and it still matches the BAM (size, mtime,
    FLAG filter): only the BGZF blocks around the candidate regions are inflated; otherwise the whole file is read."""
    if ir_backend == "einverted":
        from . import einverted
        einverted.find_einverted()          # fail before any work when the executable is missing
    import torch
    from .bamio import BamReader, read_fasta
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    params = _lib.make_params(max_direct_repeat_length=max_direct_repeat_length,
                              min_alignment_quality=min_alignment_quality,
                              min_alignment_inner_length=min_alignment_inner_length,
                              max_junction_spanning_prop=max_junction_spanning_prop,
                              large_insertion_cutoff=large_insertion_cutoff)
    flanks = pd.read_csv(flanksfile, sep="\t")                                 # pairflanks.py:23 (default NA parsing)
    gen = read_fasta(genome)
    with BamReader(bamfile) as bam:
        contig_names = [n for n, _ in bam.contigs]
        if flanks.shape[0] == 0:
            return pairflanks_packed(binding, None, contig_names, params, flanks, None, None, output_file, verbose, ir_backend)
        tab, _ = flanks_to_table(flanks, contig_names)
        g, goff = genome_arrays(gen, contig_names)
        glen = np.diff(goff)
        regions = candidate_regions(tab, max_direct_repeat_length, glen)
        # only reads that can overlap a candidate region are packed; reads over a direct-repeat window carry bases
        if decoder == "native":
            packed = pack_bam(bamfile, pool="regions", regions=regions, exclude_flags=exclude_flags, index=index_path(flanksfile, record_index))
            if verbose:
                st = packed.io_stats
                log("BAM decode (%s): %d reads over %d candidate regions, %.1f MB inflated in %.3f s" % (
                    "record index" if st["indexed"] else "whole file", packed.n_reads, len(regions), st["uncompressed_bytes"] / 1e6, st["seconds_total"]))
        elif decoder == "python":
            packed = _pack_regions_python(bam, contig_names, regions, exclude_flags)
        else:
            raise ValueError("decoder must be 'native' or 'python', not %r" % (decoder,))
    rs = device_reads(binding, packed, device)
    gdev = torch.from_numpy(g).to(device)
    return pairflanks_packed(binding, rs, contig_names, params, flanks, gdev, goff, output_file, verbose, ir_backend)

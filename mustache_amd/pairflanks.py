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
    if len(flanks) == 0:
        return tab, b""
    contigs = [str(c) for c in flanks["contig"].tolist()]
    ids = [index.get(c, -1) for c in contigs]
    if -1 in ids:
        raise KeyError("contig %r of the flanks file is not in the BAM header" % contigs[ids.index(-1)])
    tab["contig_id"] = ids
    def ints(c):                # an integer column as it is; anything else value by value, so that a NaN still raises
        v = flanks[c].to_numpy()
        return v if v.dtype.kind in "iu" else [int(x) for x in v.tolist()]
    tab["pos"] = ints("pos")
    orient = flanks["orient"].to_numpy(dtype=object)
    tab["orient"] = np.where(orient == "5p", 5, np.where(orient == "3p", 3, 0))
    for c in ("softclip_count_5p", "softclip_count_3p", "runthrough_count", "small_insertion_count_5p",
              "small_insertion_count_3p", "large_insertion_count_5p", "large_insertion_count_3p", "deletion_count",
              "upstream_deletion_count", "downstream_deletion_count", "total_count", "consensus_softclip_count"):
        tab[c] = ints(c)
    seqs = [str(s).encode("ascii") for s in flanks["consensus_seq"].tolist()]
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    tab["seq_len"] = lens
    tab["seq_off"] = np.concatenate([[0], np.cumsum(lens)[:-1]])
    return tab, b"".join(seqs)


def candidate_regions(tab, max_direct_repeat_length, genome_len):
    """Per contig, the reference intervals pairflanks can touch: the spanning fetch [pos3-1, pos5+2) and the direct
    repeat window (pairflanks.py:294-305) of every (5p, 3p) candidate.  Returns (contig_id, start, end) arrays."""
    five = tab[tab["orient"] == 5]
    three = tab[tab["orient"] == 3]
    if len(five) == 0 or len(three) == 0:
        return np.zeros((0, 5), dtype=np.int32)
    # 3p flanks sorted by (contig, position): the partners of a 5p flank are one stretch of them
    key3 = (three["contig_id"].astype(np.int64) << 33) + (three["pos"].astype(np.int64) + 2)
    order = np.argsort(key3, kind="stable")
    key3 = key3[order]
    c5, p5 = five["contig_id"].astype(np.int64), five["pos"].astype(np.int64)
    lo = np.searchsorted(key3, (c5 << 33) + np.maximum(p5 - max_direct_repeat_length - 1, -2) + 2, side="left")
    hi = np.searchsorted(key3, (c5 << 33) + p5 + 2, side="left")
    n = np.maximum(hi - lo, 0)
    a = np.repeat(np.arange(len(five)), n)                              # 5p flank of every candidate
    b = order[np.repeat(lo, n) + (np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n))]
    start, end, cid = three["pos"].astype(np.int64)[b], p5[a], c5[a]
    center = np.rint((end + start) / 2).astype(np.int64)               # Python's round(): half to even (quirk P10)
    es = np.maximum(center - TARGET_REGION_SIZE // 2, 0)
    ee = np.minimum(center + TARGET_REGION_SIZE // 2, np.asarray(genome_len, dtype=np.int64)[cid])
    return np.stack([cid, np.minimum(es, start - 1), np.maximum(ee, end + 2), es, ee], 1).astype(np.int32)


def pair_frame(result, flank_tab, flank_seqs, contig_names):
    pr, strs = result["pairs"], result["strs"]
    if len(pr) == 0:
        return pd.DataFrame({c: [] for c in PAIR_COLUMNS}, columns=PAIR_COLUMNS)
    text, ftext = bytes(strs).decode("ascii"), bytes(flank_seqs).decode("ascii")

    def col(off, n):            # None for a negative length
        return [None if m < 0 else text[o:o + m] for o, m in zip(pr[off].tolist(), pr[n].tolist())]

    def fcol(idx):
        f = flank_tab[pr[idx]]
        return [ftext[o:o + m] for o, m in zip(f["seq_off"].tolist(), f["seq_len"].tolist())]
    data = {"pair_id": np.arange(1, len(pr) + 1, dtype=np.int64),          # pairflanks.py:120
            "contig": np.asarray(contig_names, dtype=object)[pr["contig_id"]]}
    for c in ("pos_5p", "pos_3p", "softclip_count_5p", "softclip_count_3p", "total_count_5p", "total_count_3p", "spanning_count"):
        data[c] = pr[c].astype(np.int64)
    data["has_IR"] = pr["has_ir"].astype(bool)
    data["IR_length"] = pr["ir_length"].astype(np.int64)
    data["IR_5p"], data["IR_3p"] = col("ir5_off", "ir5_len"), col("ir3_off", "ir3_len")
    data["seq_5p"], data["seq_3p"] = fcol("index_5p"), fcol("index_3p")
    data["direct_repeat_reference"] = col("dr_ref_off", "dr_ref_len")
    data["direct_repeat_reads_consensus"] = col("dr_reads_off", "dr_len")
    return pd.DataFrame(data, columns=PAIR_COLUMNS)


def genome_arrays(genome, contig_names):
    """Concatenate the FASTA records in BAM contig order -> (uint8 array, int64 offsets).  A contig missing from the
    FASTA gets length 0 (the reference raises KeyError only if a pair lands on it; so do we, see pairflanks_packed)."""
    chunks, off = [], [0]
    for n in contig_names:
        s = genome.get(n, "")
        a = s if isinstance(s, np.ndarray) else np.frombuffer(s.encode("ascii"), dtype=np.uint8)   # (read_fasta_arrays gives bytes already)
        chunks.append(a)
        off.append(off[-1] + a.shape[0])
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
                      output_file=None, verbose=True, ir_backend="builtin", ir_detector=None, table=None):
    """``_pairflanks`` after the inputs are on the device.  ``flanks`` is the frame read from the findflanks TSV
    (``table``: its ``flanks_to_table`` when the caller has made it already).
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
    tab, seqs = table if table is not None else flanks_to_table(flanks, contig_names)
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
    from .bamio import BamReader, read_fasta_arrays
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    params = _lib.make_params(max_direct_repeat_length=max_direct_repeat_length,
                              min_alignment_quality=min_alignment_quality,
                              min_alignment_inner_length=min_alignment_inner_length,
                              max_junction_spanning_prop=max_junction_spanning_prop,
                              large_insertion_cutoff=large_insertion_cutoff)
    import time
    t0 = time.perf_counter()
    # the FASTA is read on a thread of its own while the flank table is parsed and the BAM's regions are decoded
    import threading
    box = {}

    def fasta():
        try:
            box["gen"] = read_fasta_arrays(genome)
        except BaseException as e:      # noqa: BLE001 -- raised by the calling thread below
            box["err"] = e
    th = threading.Thread(target=fasta)
    th.start()

    def genome_of(contig_names):
        th.join()
        if "err" in box:
            raise box["err"]
        return genome_arrays(box["gen"], contig_names)
    try:
        flanks = pd.read_csv(flanksfile, sep="\t")                             # pairflanks.py:23 (default NA parsing)
        bam = BamReader(bamfile)
    except BaseException:
        th.join()
        raise
    with bam:
        contig_names = [n for n, _ in bam.contigs]
        if flanks.shape[0] == 0:
            th.join()
            return pairflanks_packed(binding, None, contig_names, params, flanks, None, None, output_file, verbose, ir_backend)
        try:
            table = flanks_to_table(flanks, contig_names)
        except BaseException:
            th.join()
            raise
        tab = table[0]
        # Which reads are loaded is decided with the contig lengths of the BAM header (the reference clamps the direct-repeat
        # window at len(genome[contig]), pairflanks.py:294-305; the kernel does, with the FASTA's offsets): a header length that
        # differs from the FASTA's can only make a region larger, i.e. load reads that no window then uses.
        if decoder == "native":
            blen = np.asarray([max(int(n), 0) for _, n in bam.contigs], dtype=np.int64)
            regions = candidate_regions(tab, max_direct_repeat_length, blen)
            g = goff = None
        else:
            g, goff = genome_of(contig_names)
            regions = candidate_regions(tab, max_direct_repeat_length, np.diff(goff))
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
    if g is None:
        g, goff = genome_of(contig_names)
    t1 = time.perf_counter()
    rs = device_reads(binding, packed, device, clip_rec=False)
    gdev = torch.from_numpy(g).to(device)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = pairflanks_packed(binding, rs, contig_names, params, flanks, gdev, goff, output_file, verbose, ir_backend, table=table)
    t3 = time.perf_counter()
    st = getattr(packed, "io_stats", None) or {}
    from . import findflanks as _ff
    _ff.LAST_TIMING["pairflanks"] = {"tables_fasta_decode_s": round(t1 - t0, 4), "bam_decode_s": round(st.get("seconds_total", 0.0), 4),
                                     "indexed": bool(st.get("indexed", False)), "inflated_bytes": int(st.get("uncompressed_bytes", 0)),
                                     "reads_packed": int(packed.n_reads), "upload_s": round(t2 - t1, 4),
                                     "kernels_and_tsv_s": round(t3 - t2, 4), "device_ms": round(binding.device_ms(), 3)}
    return out

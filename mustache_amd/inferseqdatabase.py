"""Host side of ``mustache inferseq-database``: same function name, arguments and TSV as the reference's
``_inferseq_database`` (reference mustache/inferseqdatabase.py:28-78).  The aligner stays external exactly as in the
reference (bowtie2 --all --score-min G,1,5, end to end, through the shell: bowtie2tools.py:35-65 with the flags of
inferseqdatabase.py:47-51); what happens to its alignments afterwards -- read prefilter, couples of flank-1 / flank-2
alignments per database element, both-ends test, best summed alignment score -- runs in libmfk.so
(``mfk_inferseq_db_pairs``); the inferred sequences and the table are a few lines of host code.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

import pandas as pd

from . import lib as _lib
from .findflanks import log
from .inferseq import (INFERSEQ_COLUMNS, _NA_VALUES, alignment_table, get_flanks, revcomp, write_flanks_to_unpaired_fasta)
from .records import CSOFT_CLIP


def inferred_sequence(read1, read2, database):
    """get_inferred_sequences with add_softclipped_bases=True, inferseqdatabase.py:81-98: read1 = the forward-strand
    alignment, read2 = the reverse-strand one."""
    seq = str(database[read1.reference_name])[read1.reference_start:read2.reference_end]
    ct1, ct2 = read1.cigartuples, read2.cigartuples
    left = read1.query_sequence[:ct1[0][1]] if ct1[0][0] == CSOFT_CLIP else ""          # left_softclipped_sequence_strict
    right = read2.query_sequence[-ct2[-1][1]:] if ct2[-1][0] == CSOFT_CLIP else ""       # right_softclipped_sequence_strict
    seq = left + seq + right
    if read1.is_read2:              # the SAM flag 0x80 (never set on bowtie2 -U alignments), inferseqdatabase.py:93
        seq = revcomp(seq)
    return seq


def infer_sequences_frame(binding, reads, database, min_perc_identity, max_internal_softclip_prop, max_edge_distance):
    """infer_sequences_database after the aligner (inferseqdatabase.py:116-139) + make_dataframe (101-113) + the table
    steps of _inferseq_database (58-64): stable sort by pair id, pair id as text, empty sequences dropped."""
    reads = list(reads)
    for r in reads:
        if len(r.query_name.split("_")) != 2:     # pair_id, flank_id = read.query_name.split('_'), inferseqdatabase.py:176
            raise ValueError("not enough values to unpack: query name %r is not <pair_id>_<flank_id>" % r.query_name)
    out = {c: [] for c in INFERSEQ_COLUMNS}
    if reads:
        fwd, rev = binding.run_inferseq_db_pairs(alignment_table(reads, database), min_perc_identity,
                                                 max_internal_softclip_prop, max_edge_distance)
        for f, r in zip(fwd, rev):
            fr, rr = reads[int(f)], reads[int(r)]
            seq = inferred_sequence(fr, rr, database)
            out["pair_id"].append(int(fr.query_name.split("_")[0]))
            out["method"].append("inferred_database")
            out["loc"].append("%s:%d-%d" % (fr.reference_name, fr.reference_start, rr.reference_end))
            out["inferred_seq_length"].append(len(seq))
            out["inferred_seq"].append(seq)
    df = pd.DataFrame(out, columns=INFERSEQ_COLUMNS)
    # the library lists the couples by dense group number = order of first appearance of the pair id in the BAM; the
    # reference's dict is keyed in the order of the first KEPT read.  Either way the next line orders by pair id, stably.
    df = df.sort_values(by=["pair_id", "method"], kind="stable")
    df["pair_id"] = [str(int(x)) for x in df["pair_id"]]
    return df[df["inferred_seq_length"] > 0].reset_index(drop=True)


def align_flanks(fasta, database_path, out_bam):
    """bowtie2tools.align_fasta_to_genome (35-65) with the flags of inferseqdatabase.py:47-51; needs bowtie2 and samtools."""
    for exe in ("bowtie2", "bowtie2-build", "samtools"):
        if shutil.which(exe) is None:
            raise RuntimeError("%s is not on PATH: inferseq-database aligns the flanks with bowtie2 exactly as the "
                               "reference does; pass alignments_bam= to use alignments made elsewhere" % exe)
    if not os.path.exists(database_path + ".1.bt2"):              # index_database, inferseqdatabase.py:321-326
        subprocess.check_call(["bowtie2-build", "-o", "3", "-q", database_path, database_path])
    sam = out_bam + ".sam"
    with open(os.devnull, "w") as null:
        subprocess.check_call(["bowtie2", "--all", "--score-min", "G,1,5", "-x", database_path, "-p", "1", "-f", "-U", fasta, "-S", sam], stderr=null)
        with open(sam + ".tmp", "w") as fh:
            subprocess.check_call(["samtools", "view", "-h", "-q", "1", sam], stdout=fh, stderr=null)
        with open(out_bam, "wb") as fh:
            subprocess.check_call(["samtools", "sort", sam + ".tmp"], stdout=fh, stderr=null)
    os.remove(sam)
    os.remove(sam + ".tmp")
    return out_bam


def _inferseq_database(pairsfile, inferseq_database, min_perc_identity, max_internal_softclip_prop, max_edge_distance,
                       output_file, keep_intermediate, device="cuda:0", verbose=True, alignments_bam=None):
    """Drop-in for reference inferseqdatabase.py:28-78 (same positional arguments).  ``alignments_bam``: flank alignments
    made beforehand (a BAM with NM and AS tags, as bowtie2 writes them) instead of running bowtie2 here."""
    from .bamio import BamReader, read_fasta
    pairs = pd.read_csv(pairsfile, sep="\t", keep_default_na=False, na_values=_NA_VALUES)
    if pairs.shape[0] == 0:                                       # handle_empty_pairsfile, inferseqdatabase.py:353-362
        if not output_file:
            output_file = "mustache.inferseq_database.tsv"
        pd.DataFrame(columns=INFERSEQ_COLUMNS).to_csv(output_file, sep="\t", index=False)
        if verbose:
            log("Empty pairs file, exiting...")
        sys.exit()
    if not output_file:
        output_file = "mustache.inferseq_database.tsv"
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    database = read_fasta(inferseq_database)
    tmp = None
    if alignments_bam is None:
        tmp = tempfile.mkdtemp(prefix="mustache.inferseq_database.", dir=os.path.dirname(os.path.abspath(output_file)))
        fasta = os.path.join(tmp, "flanks.fasta")
        write_flanks_to_unpaired_fasta(get_flanks(pairs), fasta)
        if verbose:
            log("Aligning pairs to database...")
        alignments_bam = align_flanks(fasta, inferseq_database, os.path.join(tmp, "flanks.bam"))
    with BamReader(alignments_bam) as bam:
        reads = [r for r in bam if r.cigartuples is not None]
    if tmp is not None and not keep_intermediate:
        shutil.rmtree(tmp, ignore_errors=True)
    if verbose:
        log("Inferring sequences from pairs aligned to database...")
    df = infer_sequences_frame(binding, reads, database, min_perc_identity, max_internal_softclip_prop, max_edge_distance)
    if verbose:
        log("Writing results to file %s..." % output_file)
    df.insert(0, "sample", list(pairs["sample"])[0])
    df.to_csv(output_file, sep="\t", index=False)
    return df

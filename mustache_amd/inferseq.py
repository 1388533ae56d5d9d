"""Host side of ``mustache inferseq-reference``: same function name, arguments and TSV as the reference's
``_inferseq_reference`` (reference mustache/inferseqreference.py:20-52) and the ``InferSequence`` class it drives
(mustache/inferseq.py:16-183).  The aligner stays external exactly as in the reference (bowtie2 --local --all, run
through the shell, bowtie2tools.py:35-65); what happens to its alignments afterwards -- read prefilter, forward/reverse
matching, the four pair filters -- runs in libmfk.so (``mfk_inferseq_pairs``); the inferred sequences and the table are a
few lines of host code.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pandas as pd

from . import lib as _lib
from .findflanks import log
from .records import CSOFT_CLIP

INFERSEQ_COLUMNS = ["pair_id", "method", "loc", "inferred_seq_length", "inferred_seq"]
_NA_VALUES = ["-1.#IND", "1.#QNAN", "1.#IND", "-1.#QNAN", "#N/A", "N/A", "#NA", "NULL", "NaN", "-NaN", "nan", "-nan"]
_COMP = {"A": "T", "T": "A", "C": "G", "G": "C"}


def revcomp(seq):
    """misc.py:8-22: upper-cases, complements ACGT, leaves everything else as it is."""
    return "".join(_COMP.get(c, c) for c in reversed(seq.upper()))


def get_flanks(pairs):
    """InferSequence.get_flanks, inferseq.py:93-101: what is written to the FASTA the aligner sees."""
    out = []
    for _, p in pairs.iterrows():
        out.append({"pair_id": str(p["pair_id"]), "seq_5p": p["seq_5p"].upper().rstrip("N"),
                    "seq_3p": p["seq_3p"].upper().lstrip("N")})
    return out


def write_flanks_to_unpaired_fasta(flanks, path):
    """fastatools.py:72-94: all 5p flanks as <pair_id>_1, then all 3p flanks reverse-complemented as <pair_id>_2."""
    with open(path, "w") as fh:
        for p in flanks:
            fh.write(">%s_1 %s_1_5p\n%s\n" % (p["pair_id"], p["pair_id"], p["seq_5p"]))
        for p in flanks:
            fh.write(">%s_2 %s_2_3p\n%s\n" % (p["pair_id"], p["pair_id"], revcomp(p["seq_3p"])))


def alignment_table(reads, genome):
    """Alignments (pysam-like records, BAM order) -> ALN_DTYPE array for mfk_inferseq_pairs."""
    tab = np.zeros(len(reads), dtype=_lib.ALN_DTYPE)
    groups, names, contigs = {}, {}, {}
    for k, r in enumerate(reads):
        t = tab[k]
        parts = r.query_name.split("_")
        t["group"] = groups.setdefault(parts[0], len(groups))
        t["name"] = names.setdefault(r.query_name, len(names))
        t["mate"] = 1 if parts[-1] == "1" else 2
        t["is_reverse"] = 1 if r.is_reverse else 0
        t["contig_id"] = contigs.setdefault(r.reference_name, len(contigs))
        t["ref_start"], t["ref_end"] = r.reference_start, r.reference_end
        t["contig_len"] = len(genome[r.reference_name])
        t["qlen"] = len(r.query_sequence)
        ct = r.cigartuples
        t["left_clip"] = ct[0][1] if ct[0][0] == CSOFT_CLIP else 0
        t["right_clip"] = ct[-1][1] if ct[-1][0] == CSOFT_CLIP else 0
        t["aln_len"] = r.query_alignment_length
        t["nm"], t["as"] = int(r.get_tag("NM")), int(r.get_tag("AS"))
    return tab


def inferred_sequence(forward, reverse, genome):
    """InferSequence.get_inferred_sequence, inferseq.py:148-158."""
    seq = "".join(genome[forward.reference_name][forward.reference_start:reverse.reference_end])
    ct_f, ct_r = forward.cigartuples, reverse.cigartuples
    left = forward.query_sequence[:ct_f[0][1]] if ct_f[0][0] == CSOFT_CLIP else ""
    right = reverse.query_sequence[-ct_r[-1][1]:] if ct_r[-1][0] == CSOFT_CLIP else ""
    seq = left + seq + right
    if forward.query_name.split("_")[-1] == "2":                  # AlignedPair.is_reverse, inferseq.py:431-432
        seq = revcomp(seq)
    return seq


def infer_sequences_frame(binding, reads, genome, min_perc_identity, max_internal_softclip_prop, max_inferseq_size,
                          min_inferseq_size, method_name="inferred_reference"):
    """InferSequence.infer_sequences after the aligner (inferseq.py:56-70) + make_dataframe (128-146)."""
    reads = list(reads)
    out = {c: [] for c in INFERSEQ_COLUMNS}
    if reads:
        fwd, rev = binding.run_inferseq_pairs(alignment_table(reads, genome), min_perc_identity,
                                              max_internal_softclip_prop, max_inferseq_size, min_inferseq_size)
        for f, r in zip(fwd, rev):
            fr, rr = reads[int(f)], reads[int(r)]
            seq = inferred_sequence(fr, rr, genome)
            out["pair_id"].append(int(fr.query_name.split("_")[0]))
            out["method"].append(method_name)
            out["loc"].append("%s:%d-%d" % (fr.reference_name, fr.reference_start, rr.reference_end))
            out["inferred_seq_length"].append(len(seq))
            out["inferred_seq"].append(seq)
    return pd.DataFrame(out, columns=INFERSEQ_COLUMNS).sort_values(["pair_id", "method", "loc"]).reset_index(drop=True)


def align_flanks(fasta, genome_path, out_bam):
    """bowtie2tools.align_fasta_to_genome (35-65) with the flags of inferseq.py:83-87; needs bowtie2 and samtools."""
    for exe in ("bowtie2", "bowtie2-build", "samtools"):
        if shutil.which(exe) is None:
            raise RuntimeError("%s is not on PATH: inferseq-reference aligns the flanks with bowtie2 exactly as the "
                               "reference does; pass alignments_bam= to use alignments made elsewhere" % exe)
    if not os.path.exists(genome_path + ".1.bt2"):                # bowtie2tools.genome_is_indexed / index_genome
        subprocess.check_call(["bowtie2-build", "-o", "3", "-q", genome_path, genome_path])
    sam = out_bam + ".sam"
    # the reference's pipeline (bowtie2tools.py:46-60) as argument lists: no shell, so paths with spaces or shell characters
    # are safe, and a failing step raises here instead of leaving a broken BAM for the reader
    with open(os.devnull, "w") as null:
        subprocess.check_call(["bowtie2", "--all", "--score-min", "G,1,9", "--local", "-x", genome_path, "-p", "1", "-f", "-U", fasta, "-S", sam], stderr=null)
        with open(sam + ".tmp", "w") as fh:
            subprocess.check_call(["samtools", "view", "-h", "-q", "1", sam], stdout=fh, stderr=null)
        with open(out_bam, "wb") as fh:
            subprocess.check_call(["samtools", "sort", sam + ".tmp"], stdout=fh, stderr=null)
    os.remove(sam)
    os.remove(sam + ".tmp")
    return out_bam


def _inferseq_reference(pairsfile, inferseq_reference, min_perc_identity, max_internal_softclip_prop,
                        max_inferseq_size, min_inferseq_size, keep_intermediate, output_file, device="cuda:0",
                        verbose=True, alignments_bam=None):
    """Drop-in for reference inferseqreference.py:20-52 (same positional arguments).  ``alignments_bam``: flank
    alignments made beforehand (a BAM with NM and AS tags, as bowtie2 writes them) instead of running bowtie2 here."""
    from .bamio import BamReader, read_fasta
    pairs = pd.read_csv(pairsfile, sep="\t", keep_default_na=False, na_values=_NA_VALUES)
    if pairs.shape[0] == 0:                                        # handle_empty_pairsfile, inferseqreference.py:62-72
        if not output_file:
            output_file = "mustache.inferseq_reference.tsv"
        pd.DataFrame(columns=INFERSEQ_COLUMNS).to_csv(output_file, sep="\t", index=False)
        if verbose:
            log("Empty pairs file, exiting...")
        sys.exit()
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    genome = read_fasta(inferseq_reference)
    tmp = None
    if alignments_bam is None:
        tmp = tempfile.mkdtemp(prefix="mustache.inferseq.", dir=os.path.dirname(os.path.abspath(output_file)))
        fasta = os.path.join(tmp, "flanks.fasta")
        write_flanks_to_unpaired_fasta(get_flanks(pairs), fasta)
        if verbose:
            log("Aligning pairs to genome...")
        alignments_bam = align_flanks(fasta, inferseq_reference, os.path.join(tmp, "flanks.bam"))
    with BamReader(alignments_bam) as bam:
        reads = [r for r in bam if r.cigartuples is not None]
    if tmp is not None and not keep_intermediate:
        shutil.rmtree(tmp, ignore_errors=True)
    df = infer_sequences_frame(binding, reads, genome, min_perc_identity, max_internal_softclip_prop,
                               max_inferseq_size, min_inferseq_size)
    if verbose:
        log("Inferred sequences for %d pairs..." % len(set(df["pair_id"])))
        log("Writing results to file %s..." % output_file)
    df.insert(0, "sample", list(pairs["sample"])[0])
    df.to_csv(output_file, sep="\t", index=False)
    return df

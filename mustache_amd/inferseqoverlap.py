"""Host side of ``mustache inferseq-overlap``: same function name, arguments and TSV as the reference's
``_inferseq_overlap`` (reference mustache/inferseqoverlap.py:21-42).  The sliding alignment of every pair's two flank
sequences runs in libmfk.so (``mfk_overlap``); acceptance, merging and the table are a few lines of host code.
"""
from __future__ import annotations

import sys

import pandas as pd

from . import lib as _lib
from .findflanks import log

OVERLAP_COLUMNS = ["pair_id", "method", "loc", "inferred_seq_length", "inferred_seq"]
_NA_VALUES = ["-1.#IND", "1.#QNAN", "1.#IND", "-1.#QNAN", "#N/A", "N/A", "#NA", "NULL", "NaN", "-NaN", "nan", "-nan"]


def merge_overlapping_sequences(start_seq, overlap_seq1, overlap_seq2, end_seq):
    """inferseqoverlap.py:177-189: the first half of the merged sequence follows the 5p flank, the rest the 3p flank."""
    merged_length = len(start_seq) + len(overlap_seq1) + len(end_seq)
    merged = list(start_seq)
    for i in range(len(overlap_seq1)):
        merged.append(overlap_seq2[i] if len(merged) > merged_length / 2 else overlap_seq1[i])
    return "".join(merged) + end_seq


def infer_sequences_overlap(binding, pairs, min_overlap_score, min_overlap_perc_identity):
    """inferseqoverlap.py:61-91 with get_best_sliding_alignment on the device.  dict pair_id -> [(loc, length, seq)]."""
    seq5, seq3 = [str(s) for s in pairs["seq_5p"]], [str(s) for s in pairs["seq_3p"]]
    res = binding.run_overlap(seq5, seq3) if len(seq5) else []
    sequences = {}
    for pair_id, s5, s3, row in zip(pairs["pair_id"], seq5, seq3, res):
        best_score, mismatches, r_start, r_end, q_start, q_end = (int(x) for x in row)
        align_length = r_end - r_start
        if best_score >= min_overlap_score and (1 - mismatches / float(align_length)) > min_overlap_perc_identity:
            merged = merge_overlapping_sequences(s5[:q_start], s5[q_start:q_end], s3[r_start:r_end], s3[r_end:])
            loc = "seq_5p:%d-%d;seq_3p:%d-%d" % (q_start, q_end, r_start, r_end)
            sequences.setdefault(pair_id, []).append((loc, len(merged), merged))
    return sequences


def make_dataframe(inferred_sequences, method=None):
    """inferseqoverlap.py:45-58"""
    out = {c: [] for c in OVERLAP_COLUMNS}
    for pair_id, results in inferred_sequences.items():
        for loc, length, seq in results:
            out["pair_id"].append(pair_id)
            out["method"].append(method)
            out["loc"].append(str(loc))
            out["inferred_seq_length"].append(length)
            out["inferred_seq"].append(seq)
    return pd.DataFrame(out, columns=OVERLAP_COLUMNS)


def inferseq_overlap_frame(binding, pairs, min_overlap_score, min_overlap_perc_identity, min_inferseq_size):
    sequences = infer_sequences_overlap(binding, pairs, min_overlap_score, min_overlap_perc_identity)
    method1 = make_dataframe(sequences, method="inferred_overlap")
    method1["pair_id"] = [str(int(p)) for p in method1["pair_id"]]                  # inferseqoverlap.py:32
    method1 = method1[method1["inferred_seq_length"] >= min_inferseq_size].copy()     # inferseqoverlap.py:33
    method1.insert(0, "sample", list(pairs["sample"])[0] if pairs.shape[0] > 0 else None)
    return method1


def _inferseq_overlap(pairsfile, min_overlap_score, min_overlap_perc_identity, min_inferseq_size, output_file,
                      device="cuda:0", verbose=True):
    """Drop-in for reference inferseqoverlap.py:21-42 (same positional arguments)."""
    pairs = pd.read_csv(pairsfile, sep="\t", keep_default_na=False, na_values=_NA_VALUES)
    if pairs.shape[0] == 0:                                                            # handle_empty_pairsfile, 166-175
        if not output_file:
            output_file = "mustache.inferseq_overlap.tsv"
        pd.DataFrame(columns=OVERLAP_COLUMNS).to_csv(output_file, sep="\t", index=False)
        if verbose:
            log("Empty pairs file, exiting...")
        sys.exit()
    binding = _lib.load(int(str(device).split(":")[1]) if ":" in str(device) else 0)
    if verbose:
        log("Inferring sequences from overlap...")
    method1 = inferseq_overlap_frame(binding, pairs, min_overlap_score, min_overlap_perc_identity, min_inferseq_size)
    if verbose:
        log("Writing results to file %s..." % output_file)
    method1.to_csv(output_file, sep="\t", index=False)
    return method1

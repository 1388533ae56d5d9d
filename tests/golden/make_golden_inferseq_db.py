"""Generates tests/golden/inferseq_db_vectors.json.gz by RUNNING THE REFERENCE'S infer_sequences_database (build container only).

    python tests/golden/make_golden_inferseq_db.py

bowtie2 is not in this image, so the flank alignments against the element database are synthetic: per pair id, on one or
several database elements, alignments of flank 1 near one end and of flank 2 near the other (both orientations), plus the
situations the code reacts to (end-to-end alignment scores are <= 0 and keep_best_alignment_score starts from 0, so only
couples of perfect alignments survive unless a score is positive: some are): alignments too far from an edge, soft clips hanging over an element's end (kept) or inside
it (dropped by proportion), low identity, several alignments of a flank on one element (the greedy match_pairs, with its
read1.reference_start on line 284), both flanks on one strand, elements with only one flank, ties of the summed alignment
score, negative scores, forward reads that start behind the reverse read's end (empty slice).  The reference's own code
(inferseqdatabase.py:116-322 and the table steps 56-64) turns them into the table; stored: the alignments, the database,
the parameters and the TSV text.  Data only.
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from mustache_amd.records import AlignedRead  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def make_read(rng, name, contig, start, alen, reverse, left_clip=0, right_clip=0, nm=0, as_=None, flag_extra=0):
    cig = ([(4, left_clip)] if left_clip else []) + [(0, alen)] + ([(4, right_clip)] if right_clip else [])
    seq = rand_seq(rng, left_clip + alen + right_clip)
    return AlignedRead(name, contig, start, 40, cig, seq, [30] * len(seq), md=str(alen), flag=(16 if reverse else 0) | flag_extra,
                       tags={"NM": nm, "AS": int(as_ if as_ is not None else -6 * nm - (int(rng.integers(1, 4)) if rng.random() < 0.25 else 0))})


def make_case(rng, n_pairs, L):
    db = {"IS%d" % k: rand_seq(rng, int(rng.integers(120, 900))) for k in range(int(rng.integers(2, 6)))}
    names = list(db)
    reads = []
    for pid in range(1, n_pairs + 1):
        pid_s = str(pid * int(rng.integers(1, 4)) + 50 * int(rng.integers(0, 3)))
        if any(r.query_name.split("_")[0] == pid_s for r in reads):
            continue
        for el in rng.permutation(len(names))[:int(rng.integers(1, 4))]:
            contig = names[int(el)]
            clen = len(db[contig])
            flipped = rng.random() < 0.4
            five, three = ("2", "1") if flipped else ("1", "2")
            kind = int(rng.integers(0, 12))
            for _ in range(int(rng.integers(1, 4))):                 # forward alignments near the element's start
                alen = int(rng.integers(15, 40))
                start = int(rng.integers(0, L + 3))                  # some beyond max_edge_distance
                lc = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
                rc = int(rng.integers(1, 6)) if kind == 0 else 0     # internal right clip: dropped by proportion
                reads.append(make_read(rng, pid_s + "_" + five, contig, start, min(alen, clen - start), False, left_clip=lc, right_clip=rc,
                                       nm=3 if kind == 1 else (int(rng.integers(0, 2)) if rng.random() < 0.3 else 0), flag_extra=0x80 if kind == 2 else 0))
            if kind == 3:
                continue                                              # only one flank on this element
            for _ in range(int(rng.integers(1, 4))):                 # reverse alignments near the element's end
                alen = int(rng.integers(15, 40))
                end = clen - int(rng.integers(0, L + 3))
                rc = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
                lc = int(rng.integers(1, 6)) if kind == 4 else 0     # internal left clip
                reads.append(make_read(rng, pid_s + "_" + three, contig, max(0, end - alen), min(alen, end), True, left_clip=lc, right_clip=rc,
                                       nm=int(rng.integers(0, 2)) if rng.random() < 0.3 else 0, as_=int(rng.integers(1, 30)) if kind == 5 else None))
            if kind == 6:                                             # the other flank on the same strand as well
                reads.append(make_read(rng, pid_s + "_" + three, contig, int(rng.integers(0, L + 1)), 20, False))
            if kind == 7:                                             # a short element: the forward read starts behind the reverse read's end
                reads.append(make_read(rng, pid_s + "_" + five, contig, min(L, clen - 16), 15, False))
    order = {c: i for i, c in enumerate(names)}
    reads.sort(key=lambda r: (order[r.reference_name], r.reference_start))   # a coordinate-sorted BAM
    return db, reads


def read_row(r):
    return [r.query_name, r.reference_name, r.reference_start, r.cigarstring, r.query_sequence, r.flag, r.tags["NM"], r.tags["AS"]]


PARAM_SETS = [dict(), dict(min_perc_identity=0.99), dict(max_internal_softclip_prop=0.0), dict(max_edge_distance=3),
              dict(min_perc_identity=0.5, max_internal_softclip_prop=0.3, max_edge_distance=25)]


def main():
    rng = np.random.default_rng(20261111)
    cases, rows_total = [], 0
    while len(cases) < 60:
        params = PARAM_SETS[len(cases) % len(PARAM_SETS)]
        db, reads = make_case(rng, int(rng.integers(1, 9)), params.get("max_edge_distance", 10))
        tsv = rh.run_inferseq_database(db, reads, **params)
        rows_total += tsv.count("\n") - 1
        cases.append({"database": db, "reads": [read_row(r) for r in reads], "params": params, "tsv": tsv})
    cases.append({"database": {"IS1": "ACGT" * 50}, "reads": [], "params": {}, "tsv": rh.run_inferseq_database({"IS1": "ACGT" * 50}, [])})
    out = os.path.join(HERE, "inferseq_db_vectors.json.gz")
    with gzip.open(out, "wt") as fh:
        json.dump({"cases": cases}, fh)
    print("wrote %s: %d cases, %d result rows, %d bytes" % (out, len(cases), rows_total, os.path.getsize(out)))


if __name__ == "__main__":
    main()

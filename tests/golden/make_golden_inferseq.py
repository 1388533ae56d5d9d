"""Generates tests/golden/inferseq_vectors.json.gz by RUNNING THE REFERENCE'S InferSequence (build container only).

    python tests/golden/make_golden_inferseq.py

bowtie2 is not in this image, so the flank alignments are synthetic: per pair id a few loci with a forward alignment of
one flank and reverse alignments of the other at several distances, plus the situations the pairing logic reacts to
(soft-clipped ends that the prefilter drops or keeps, low identity, two alignments at one position, multi-hits that share
a forward or a reverse read, both orientations, alignments at contig ends, hits on several contigs).  The reference's
own code (inferseq.py:103-183, 185-395) turns them into the table; stored: the alignments, the genome, the parameters
and the TSV text.  Data only.  Sets whose result depends on the reference's set iteration order (two candidate pairs of
equal length in one filter_smallest_overlapping group, e.g. two alignments of a flank at one position) are detected by
re-running the reference with shuffled set orders (ref_harness.run_inferseq, shuffle_seed) and are not stored.
"""
import gzip
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from mustache_amd.records import AlignedRead  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def make_read(rng, name, contig, start, alen, reverse, left_clip=0, right_clip=0, nm=0, as_=None):
    cig = ([(4, left_clip)] if left_clip else []) + [(0, alen)] + ([(4, right_clip)] if right_clip else [])
    seq = rand_seq(rng, left_clip + alen + right_clip)
    return AlignedRead(name, contig, start, 40, cig, seq, [30] * len(seq), md=str(alen), flag=16 if reverse else 0,
                       tags={"NM": nm, "AS": int(as_ if as_ is not None else 2 * alen - 6 * nm)})


def make_case(rng, n_pairs):
    genome = {"chrA": rand_seq(rng, int(rng.integers(3000, 6000))), "pB": rand_seq(rng, int(rng.integers(800, 1500)))}
    if rng.random() < 0.5:
        genome["pC"] = rand_seq(rng, 400)
    contigs = list(genome)
    reads = []
    for pid in range(1, n_pairs + 1):
        pid_s = str(pid * int(rng.integers(1, 4)) + 100 * int(rng.integers(0, 3)))   # ids are not 1..n and not sorted
        if any(r.query_name.split("_")[0] == pid_s for r in reads):
            continue
        used = set()
        for _ in range(int(rng.integers(1, 4))):                     # loci
            contig = contigs[int(rng.integers(0, len(contigs)))]
            clen = len(genome[contig])
            flipped = rng.random() < 0.4                              # element inserted in the other orientation
            fmate, rmate = ("2", "1") if flipped else ("1", "2")
            a = int(rng.integers(0, max(1, clen - 300)))
            while a in used:
                a += 7
            used.add(a)
            alen = int(rng.integers(18, 45))
            kind = rng.integers(0, 10)
            lc = 0
            if kind == 0:
                lc = int(rng.integers(1, 4))                          # forward read with a left clip: dropped if > 1 and not at the contig start
            if kind == 1:
                a = 0; lc = 3                                         # left clip hanging over the contig start: kept
            nm = 3 if kind == 2 else int(rng.integers(0, 2))          # low identity: dropped
            reads.append(make_read(rng, pid_s + "_" + fmate, contig, a, alen, False, left_clip=lc,
                                   right_clip=int(rng.integers(0, 5)) if kind == 3 else 0, nm=nm))
            if kind == 4:                                             # second forward alignment at the same position (dict overwrite)
                reads.append(make_read(rng, pid_s + "_" + fmate, contig, a, alen + 2, False, as_=int(rng.integers(20, 90))))
            for _ in range(int(rng.integers(1, 4))):                  # reverse alignments of the other flank at several distances
                L = int(rng.integers(20, 1800))
                end = min(clen, a + L)
                ralen = int(rng.integers(18, 45))
                if end - ralen <= 0 or end in used:
                    continue
                used.add(end)
                rk = rng.integers(0, 8)
                rc = int(rng.integers(1, 4)) if rk == 0 else 0        # right clip: dropped if > 1 and not at the contig end
                if rk == 1:
                    end = clen; rc = 2                                # right clip at the contig end: kept
                reads.append(make_read(rng, pid_s + "_" + rmate, contig, end - ralen, ralen, True, right_clip=rc,
                                       left_clip=int(rng.integers(0, 5)) if rk == 2 else 0, nm=int(rng.integers(0, 2))))
            if rng.random() < 0.3:                                    # a second forward read further left pairing with the same reverse reads
                a2 = max(0, a - int(rng.integers(30, 200)))
                if a2 not in used:
                    used.add(a2)
                    reads.append(make_read(rng, pid_s + "_" + fmate, contig, a2, alen, False))
            if rng.random() < 0.2:                                    # same-mate reverse read: never a partner
                reads.append(make_read(rng, pid_s + "_" + fmate, contig, a + 50, 25, True))
    order = {c: i for i, c in enumerate(contigs)}
    reads.sort(key=lambda r: (order[r.reference_name], r.reference_start))   # a coordinate-sorted BAM
    return genome, reads


def read_row(r):
    return [r.query_name, r.reference_name, r.reference_start, r.cigarstring, r.query_sequence, r.flag, r.tags["NM"], r.tags["AS"]]


PARAM_SETS = [dict(), dict(min_perc_identity=0.99), dict(max_internal_softclip_prop=0.0), dict(max_inferseq_size=900, min_inferseq_size=100),
              dict(min_perc_identity=0.5, max_internal_softclip_prop=0.2, min_inferseq_size=1)]


def order_independent(genome, reads, params, want):
    """Results that change with the iteration order of the reference's sets of pair objects are no use as vectors."""
    pairs = pd.DataFrame({"pair_id": [1], "seq_5p": ["A"], "seq_3p": ["A"]})
    return all(rh.run_inferseq(pairs, genome, reads, shuffle_seed=seed, **params) == want for seed in range(1, 9))


def main():
    rng = np.random.default_rng(20260611)
    cases, rows_total = [], 0
    pairs = pd.DataFrame({"pair_id": [1], "seq_5p": ["A"], "seq_3p": ["A"]})   # InferSequence only keeps it; not used after the aligner
    while len(cases) < 40:
        genome, reads = make_case(rng, int(rng.integers(1, 9)))
        params = PARAM_SETS[len(cases) % len(PARAM_SETS)]
        tsv = rh.run_inferseq(pairs, genome, reads, **params)
        if not order_independent(genome, reads, params, tsv):
            print("skipping a set-order dependent case")
            continue
        rows_total += tsv.count("\n") - 1
        cases.append({"genome": genome, "reads": [read_row(r) for r in reads], "params": params, "tsv": tsv})
    # hand-made: nothing aligned; one lonely forward read
    cases.append({"genome": {"chrA": "ACGT" * 50}, "reads": [], "params": {}, "tsv": rh.run_inferseq(pairs, {"chrA": "ACGT" * 50}, [])})
    lone = [make_read(rng, "5_1", "chrA", 10, 30, False)]
    cases.append({"genome": {"chrA": "ACGT" * 50}, "reads": [read_row(r) for r in lone], "params": {},
                  "tsv": rh.run_inferseq(pairs, {"chrA": "ACGT" * 50}, lone)})
    out = os.path.join(HERE, "inferseq_vectors.json.gz")
    with gzip.open(out, "wt") as fh:
        json.dump({"cases": cases}, fh)
    print("wrote %s: %d cases, %d result rows, %d bytes" % (out, len(cases), rows_total, os.path.getsize(out)))


if __name__ == "__main__":
    main()

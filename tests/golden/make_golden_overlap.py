"""Generates tests/golden/overlap_vectors.json.gz by RUNNING THE REFERENCE'S inferseq-overlap (build container only).

    python tests/golden/make_golden_overlap.py

(1) unit vectors: get_best_sliding_alignment (mustache/inferseqoverlap.py:94-164) on string pairs of many shapes;
(2) end to end: _inferseq_overlap (21-42) on pairs tables -- the committed cases' own pairflanks TSVs, made-up tables
    whose flanks do overlap (clean, with mismatches, one flank inside the other, too short), several parameter sets,
    and a header-only table.  Stored: inputs, parameters, and the tuples / TSV text the reference produced.  Data only.
"""
import contextlib
import gzip
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from oracle import ref_harness as rh  # noqa: E402
from tests import util  # noqa: E402


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


def mutate(rng, s, rate):
    out = list(s)
    for i in range(len(out)):
        if rng.random() < rate:
            out[i] = "ACGT"[("ACGT".index(out[i]) + 1 + int(rng.integers(0, 3))) % 4] if out[i] in "ACGT" else "A"
    return "".join(out)


def unit_pairs(rng):
    pairs = [("A", "A"), ("A", "C"), ("AC", "A"), ("A", "CA"), ("ACGT", "ACGT"), ("ACGT", "TGCA"), ("AAAA", "AAAAAAAA"),
             ("AAAAAAAA", "AAAA"), ("ACGTN", "NACGT"), ("acgt", "ACGT")]
    for _ in range(150):
        lq, lr = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        kind = rng.integers(0, 4)
        if kind == 0:
            pairs.append((rand_seq(rng, lq), rand_seq(rng, lr)))
        elif kind == 1:                              # suffix of q = prefix of r, with noise
            ov = int(rng.integers(1, min(lq, lr) + 1))
            core = rand_seq(rng, ov)
            pairs.append((rand_seq(rng, lq - ov) + core, mutate(rng, core, 0.1) + rand_seq(rng, lr - ov)))
        elif kind == 2:                              # low-complexity: many ties
            pairs.append((rand_seq(rng, lq, "AC"), rand_seq(rng, lr, "AC")))
        else:                                        # one inside the other
            big = rand_seq(rng, max(lq, lr) + 5)
            a = int(rng.integers(0, 5))
            small = big[a:a + min(lq, lr)]
            pairs.append((big, small) if rng.random() < 0.5 else (small, big))
    return pairs


def made_up_table(rng, n_rows):
    rows = ["sample\tpair_id\tcontig\tpos_5p\tpos_3p\tseq_5p\tseq_3p"]
    for i in range(n_rows):
        ins = rand_seq(rng, int(rng.integers(25, 220)))
        a, b = int(rng.integers(10, len(ins) + 1)), int(rng.integers(10, len(ins) + 1))
        s5, s3 = ins[:a], ins[len(ins) - b:]
        kind = i % 5
        if kind == 1:
            s3 = mutate(rng, s3, 0.05)
        elif kind == 2:
            s5 = mutate(rng, s5, 0.15)
        elif kind == 3:
            s5, s3 = rand_seq(rng, a), rand_seq(rng, b)
        pid = i + 1 if i % 7 else max(i, 1)          # a repeated pair_id now and then (rows group by pair_id)
        rows.append("sampleX\t%d\tc1\t%d\t%d\t%s\t%s" % (pid, 1000 + 10 * i, 995 + 10 * i, s5, s3))
    return "\n".join(rows) + "\n"


def main():
    rng = np.random.default_rng(2024)
    sliding, _find = rh.reference_overlap_functions()
    unit = [[q, r, [int(x) for x in sliding(q, r)]] for q, r in unit_pairs(rng)]
    tables = []
    for name in util.CASES:
        tables.append((name + "_own_pairs", util.load_json(name)["pairflanks_tsv"], {}))
    tables.append(("made_up_default", made_up_table(rng, 60), {}))
    tables.append(("made_up_loose", made_up_table(rng, 60), {"min_overlap_score": 4, "min_overlap_perc_identity": 0.7, "min_inferseq_size": 1}))
    tables.append(("made_up_strict", made_up_table(rng, 40), {"min_overlap_score": 25, "min_overlap_perc_identity": 0.99, "min_inferseq_size": 100}))
    tables.append(("header_only", "sample\tpair_id\tcontig\tpos_5p\tpos_3p\tseq_5p\tseq_3p\n", {}))
    e2e = []
    for label, tsv, params in tables:
        tmp = tempfile.mkdtemp()
        pf, of = os.path.join(tmp, "pairs.tsv"), os.path.join(tmp, "out.tsv")
        with open(pf, "w") as fh:
            fh.write(tsv)
        with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
            out = rh.run_inferseq_overlap(pf, of, **params)
        e2e.append({"label": label, "pairs_tsv": tsv, "params": params, "overlap_tsv": out})
        print("%-20s rows out = %d" % (label, len(out.splitlines()) - 1))
    with gzip.open(os.path.join(HERE, "overlap_vectors.json.gz"), "wt") as fh:
        json.dump({"sliding": unit, "tables": e2e}, fh)
    print("unit vectors:", len(unit))


if __name__ == "__main__":
    main()

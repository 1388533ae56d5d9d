"""Generates tests/golden/recall_vectors.json.gz by RUNNING THE REFERENCE'S recall (build container only).

    python tests/golden/make_golden_recall.py

For every committed end-to-end case (its alignments are already in tests/golden/<case>.json.gz) the reference's
``_recall`` (mustache/recall.py, imported from /root/reference through oracle/ref_harness.py) is run twice: on the case's
own pairflanks TSV, and on a made-up pairs table whose positions include contig ends, positions without any read and
positions next to indels.  Stored: the pairs table, the parameters and the TSV the reference wrote.  Data only.
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

CASES = ["quirks", "sim_a", "sim_b", "sim_c", "sim_d"]


def made_up_pairs(case, seed):
    rng = np.random.default_rng(seed)
    rows = ["contig\tpos_5p\tpos_3p"]
    contigs = case["contigs"]
    # every contig in reverse header order (the output must follow first appearance, not the header), ends included
    for name, length in reversed(contigs):
        rows.append("%s\t%d\t%d" % (name, length - 1, 0))
        rows.append("%s\t%d\t%d" % (name, length, -1))
        for _ in range(12):
            p = int(rng.integers(0, length))
            rows.append("%s\t%d\t%d" % (name, p, max(p - int(rng.integers(0, 9)), 0)))
    # positions of the case's own findflanks rows and their neighbours (clips, indels, deletions around real sites)
    for line in case["findflanks_tsv"].splitlines()[1:]:
        f = line.split("\t")
        contig, pos = f[2], int(f[3])
        rows.append("%s\t%d\t%d" % (contig, pos + 1, max(pos - 1, -1)))
        rows.append("%s\t%d\t%d" % (contig, pos, pos))
    return "\n".join(rows) + "\n"


def main():
    out = {}
    for k, name in enumerate(CASES):
        case = util.load_json(name)
        rs = util.case_readset(case)
        entries = []
        variants = [("own_pairs", case["pairflanks_tsv"], {}),
                    ("made_up", made_up_pairs(case, 100 + k), {}),
                    ("made_up_strict", made_up_pairs(case, 200 + k),
                     {"min_alignment_quality": 30, "min_alignment_inner_length": 40, "large_insertion_cutoff": 5})]
        for label, pairs_tsv, params in variants:
            if len(pairs_tsv.splitlines()) < 2:
                continue
            tmp = tempfile.mkdtemp()
            pf, of = os.path.join(tmp, "pairs.tsv"), os.path.join(tmp, "recall.tsv")
            with open(pf, "w") as fh:
                fh.write(pairs_tsv)
            with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
                df, tsv = rh.run_recall(pf, rs, of, **params)
            entries.append({"label": label, "pairs_tsv": pairs_tsv, "params": params, "recall_tsv": tsv})
            print("%-8s %-15s sites=%d" % (name, label, len(df) // 2))
        out[name] = entries
    with gzip.open(os.path.join(HERE, "recall_vectors.json.gz"), "wt") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()

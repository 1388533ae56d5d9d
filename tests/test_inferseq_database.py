"""inferseq-database (second half): pairing of flank alignments against a database of known elements.
Fixtures = the reference's own infer_sequences_database run on 61 sets of synthetic alignments
(tests/golden/make_golden_inferseq_db.py); the oracle restatement must reproduce them (CPU), the HIP kernel must equal
both (GPU), also on dense random tables."""
import io
import os

import numpy as np
import pandas as pd
import pytest
from click.testing import CliRunner

from mustache_amd import bamio, inferseqdatabase as isd, lib
from mustache_amd.inferseq import alignment_table
from mustache_amd.main import cli, inferseq_database
from mustache_amd.records import AlignedRead, parse_cigar_string
from tests import util


def _reads(rows):
    out = []
    for name, contig, start, cigar, seq, flag, nm, as_ in rows:
        out.append(AlignedRead(name, contig, start, 40, parse_cigar_string(cigar), seq, [30] * len(seq), md="1", flag=flag,
                               tags={"NM": nm, "AS": as_}))
    return out


def _params(p):
    d = dict(min_perc_identity=0.95, max_internal_softclip_prop=0.05, max_edge_distance=10)
    d.update(p)
    return d


def _tsv(binding, case):
    p = _params(case["params"])
    df = isd.infer_sequences_frame(binding, _reads(case["reads"]), case["database"], p["min_perc_identity"],
                                   p["max_internal_softclip_prop"], p["max_edge_distance"])
    return df.to_csv(sep="\t", index=False)


def test_oracle_reproduces_the_reference_tables(oracle):
    cases = util.load_json("inferseq_db_vectors")["cases"]
    assert len(cases) >= 60 and sum(c["tsv"].count("\n") - 1 for c in cases) > 100
    for k, case in enumerate(cases):
        assert _tsv(oracle, case) == case["tsv"], k


def _random_table(rng, n, n_groups, n_contigs):
    t = np.zeros(n, dtype=lib.ALN_DTYPE)
    clen = rng.integers(60, 400, n_contigs)
    t["group"] = np.sort(rng.integers(0, n_groups, n))
    t["mate"] = rng.integers(1, 3, n)
    t["name"] = t["group"] * 2 + t["mate"] - 1
    t["is_reverse"] = rng.integers(0, 2, n)
    t["contig_id"] = rng.integers(0, n_contigs, n)
    t["contig_len"] = clen[t["contig_id"]]
    alen = rng.integers(10, 40, n)
    near = rng.integers(0, 14, n)
    t["ref_start"] = np.where(t["is_reverse"] == 1, np.maximum(0, t["contig_len"] - near - alen), near)
    t["ref_end"] = np.minimum(t["contig_len"], t["ref_start"] + alen)
    t["left_clip"] = np.where(rng.random(n) < 0.3, rng.integers(1, 5, n), 0)
    t["right_clip"] = np.where(rng.random(n) < 0.3, rng.integers(1, 5, n), 0)
    t["aln_len"] = t["ref_end"] - t["ref_start"]
    t["qlen"] = t["aln_len"] + t["left_clip"] + t["right_clip"]
    t["nm"] = np.where(rng.random(n) < 0.2, rng.integers(1, 4, n), 0)
    t["as"] = np.where(rng.random(n) < 0.6, 0, -rng.integers(1, 20, n))
    order = np.lexsort((t["ref_start"], t["contig_id"]))     # BAM order
    return t[order]


def test_oracle_on_random_tables_is_self_consistent(oracle):
    rng = np.random.default_rng(5)
    t = _random_table(rng, 400, 30, 5)
    f, r = oracle.run_inferseq_db_pairs(t, 0.9, 0.05, 10)
    assert len(f) > 0
    assert np.all(t["is_reverse"][f] == 0) and np.all(t["is_reverse"][r] == 1)
    assert np.all(t["group"][f] == t["group"][r]) and np.all(t["contig_id"][f] == t["contig_id"][r]) and np.all(t["mate"][f] != t["mate"][r])


def test_command_flags():
    opts = {p.name: p for p in inferseq_database.params}
    for name, (short, default) in {"min_perc_identity": ("-minident", 0.90), "max_internal_softclip_prop": ("-maxclip", 0.05),
                                   "max_edge_distance": ("-maxedgedist", 10), "output_file": ("-o", "mustache.inferseq_database.tsv")}.items():
        assert "--" + name in opts[name].opts and short in opts[name].opts and opts[name].default == default   # reference main.py:201-208
    assert opts["keep_intermediate"].default is False


@pytest.mark.gpu
def test_hip_equals_the_reference_tables_and_the_oracle(hip, oracle):
    for k, case in enumerate(util.load_json("inferseq_db_vectors")["cases"]):
        assert _tsv(hip, case) == case["tsv"], k
    rng = np.random.default_rng(11)
    for n, g, c in ((50, 3, 2), (400, 30, 5), (3000, 40, 6), (20000, 300, 12), (5000, 2, 1)):
        t = _random_table(rng, n, g, c)
        for params in ((0.9, 0.05, 10), (0.5, 0.5, 13), (0.99, 0.0, 2)):
            fo, ro = oracle.run_inferseq_db_pairs(t, *params)
            fh, rh = hip.run_inferseq_db_pairs(t, *params)
            assert np.array_equal(fo, fh) and np.array_equal(ro, rh), (n, params)


@pytest.mark.gpu
def test_cli_from_alignments_bam(tmp_path, hip):
    """pairs TSV + database FASTA + a BAM of flank alignments -> the command -> the reference's table with the sample column."""
    case = [c for c in util.load_json("inferseq_db_vectors")["cases"] if c["tsv"].count("\n") > 3][0]
    reads = _reads(case["reads"])
    contigs = [(n, len(s)) for n, s in case["database"].items()]
    bam, fa, pairs, out = (os.path.join(tmp_path, x) for x in ("aln.bam", "db.fa", "pairs.tsv", "out.tsv"))
    with bamio.BamWriter(bam, contigs) as w:
        for r in reads:
            w.write(r)
    bamio.write_fasta(fa, case["database"])
    pd.DataFrame({"sample": ["s1"], "pair_id": [1], "seq_5p": ["ACGT"], "seq_3p": ["ACGT"]}).to_csv(pairs, sep="\t", index=False)
    p = _params(case["params"])
    r = CliRunner().invoke(cli, ["inferseq-database", pairs, fa, "-o", out, "--alignments-bam", bam, "-minident", str(p["min_perc_identity"]),
                                 "-maxclip", str(p["max_internal_softclip_prop"]), "-maxedgedist", str(p["max_edge_distance"])], catch_exceptions=False)
    assert r.exit_code == 0, r.output
    want = pd.read_csv(io.StringIO(case["tsv"]), sep="\t", dtype=str)
    want.insert(0, "sample", "s1")
    assert open(out).read() == want.to_csv(sep="\t", index=False)

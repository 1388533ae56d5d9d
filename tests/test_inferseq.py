"""inferseq-reference pairing: mfk_inferseq_pairs against the tables the reference's InferSequence produced
from the same alignments (tests/golden/inferseq_vectors.json.gz, made by tests/golden/make_golden_inferseq.py)."""
import numpy as np
import pytest

from mustache_amd import inferseq as isq
from mustache_amd.records import AlignedRead, parse_cigar_string
from tests import util


def case_reads(case):
    return [AlignedRead(name, contig, start, 40, parse_cigar_string(cigar), seq, [30] * len(seq), flag=flag,
                        tags={"NM": nm, "AS": as_}) for name, contig, start, cigar, seq, flag, nm, as_ in case["reads"]]


def table_text(binding, case):
    p = dict(min_perc_identity=0.95, max_internal_softclip_prop=0.05, max_inferseq_size=200000, min_inferseq_size=30)
    p.update(case["params"])
    return isq.infer_sequences_frame(binding, case_reads(case), case["genome"], **p).to_csv(sep="\t", index=False)


def random_alignments(rng, n, n_groups):
    """Dense random tables: many reads per pair id on few positions, so every branch of the filters is taken often."""
    tab = np.zeros(n, dtype=isq._lib.ALN_DTYPE)
    tab["group"] = rng.integers(0, n_groups, n)
    tab["mate"] = rng.integers(1, 3, n)
    tab["name"] = tab["group"] * 2 + tab["mate"] - 1
    tab["is_reverse"] = rng.integers(0, 2, n)
    tab["contig_id"] = rng.integers(0, 2, n)
    tab["ref_start"] = rng.integers(0, 60, n) * 5
    tab["aln_len"] = rng.integers(15, 40, n)
    tab["ref_end"] = tab["ref_start"] + tab["aln_len"]
    tab["contig_len"] = 340
    tab["left_clip"] = np.where(rng.random(n) < 0.2, rng.integers(1, 6, n), 0)
    tab["right_clip"] = np.where(rng.random(n) < 0.2, rng.integers(1, 6, n), 0)
    tab["qlen"] = tab["aln_len"] + tab["left_clip"] + tab["right_clip"]
    tab["nm"] = rng.integers(0, 3, n)
    tab["as"] = rng.integers(20, 26, n)
    return tab


def test_oracle_reproduces_the_reference_tables(oracle):
    cases = util.load_json("inferseq_vectors")["cases"]
    assert len(cases) >= 40 and sum(c["tsv"].count("\n") - 1 for c in cases) >= 100
    for k, case in enumerate(cases):
        assert table_text(oracle, case) == case["tsv"], k


def test_zero_alignment_length_is_the_reference_error(oracle):
    tab = random_alignments(np.random.default_rng(1), 5, 2)
    tab["aln_len"][3] = 0
    with pytest.raises(isq._lib.MfkError, match="ZeroDivisionError"):
        oracle.run_inferseq_pairs(tab, 0.95, 0.05, 200000, 30)


@pytest.mark.gpu
def test_hip_reproduces_the_reference_tables(hip):
    for k, case in enumerate(util.load_json("inferseq_vectors")["cases"]):
        assert table_text(hip, case) == case["tsv"], k


@pytest.mark.gpu
def test_hip_matches_oracle_on_dense_random_tables(hip, oracle):
    rng = np.random.default_rng(77)
    for n, g in ((0, 1), (1, 1), (40, 1), (300, 7), (3000, 40), (1500, 1), (20000, 3000)):
        tab = random_alignments(rng, n, g)
        for params in ((0.95, 0.05, 200000, 30), (0.9, 0.2, 150, 1)):
            a, b = hip.run_inferseq_pairs(tab, *params), oracle.run_inferseq_pairs(tab, *params)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (n, g, params)
        if n >= 300:
            assert len(a[0]) > 0
    tab = random_alignments(rng, 5, 2)
    tab["aln_len"][3] = 0
    with pytest.raises(isq._lib.MfkError, match="ZeroDivisionError"):
        hip.run_inferseq_pairs(tab, 0.95, 0.05, 200000, 30)


@pytest.mark.gpu
def test_cli_from_an_alignment_bam(tmp_path):
    """pairs TSV + FASTA + a BAM of flank alignments on disk -> `inferseq-reference --alignments-bam` -> the reference's
    table with the sample column in front (inferseqreference.py:44-50)."""
    import os

    from click.testing import CliRunner
    from mustache_amd import bamio
    from mustache_amd.main import cli
    from mustache_amd.records import ReadSet
    case = util.load_json("inferseq_vectors")["cases"][3]
    reads = case_reads(case)
    contigs = [(n, len(s)) for n, s in case["genome"].items()]
    bam, fa, pairs, out = (os.path.join(tmp_path, x) for x in ("f.bam", "g.fa", "p.tsv", "o.tsv"))
    bamio.write_bam(bam, ReadSet(contigs, reads))
    bamio.write_fasta(fa, case["genome"])
    open(pairs, "w").write("sample\tpair_id\tseq_5p\tseq_3p\nS1\t1\tACGT\tACGT\n")
    args = ["inferseq-reference", pairs, fa, "-o", out, "--alignments-bam", bam]
    for k, v in case["params"].items():
        args += ["--" + k, str(v)]
    r = CliRunner().invoke(cli, args)
    assert r.exit_code == 0, r.output
    want = "\n".join(("sample" if i == 0 else "S1") + "\t" + line for i, line in enumerate(case["tsv"].splitlines())) + "\n"
    assert open(out).read() == want and want.count("\n") > 2

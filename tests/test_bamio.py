import os

from mustache_amd import bamio
from tests import util


def test_bam_round_trip(tmp_path):
    case = util.load_json("quirks")
    rs = util.case_readset(case)
    path = os.path.join(tmp_path, "q.bam")
    bamio.write_bam(path, rs)
    with bamio.BamReader(path) as bam:
        assert bam.contigs == rs.contigs
        back = list(bam)
    assert len(back) == len(rs.reads)
    for a, b in zip(rs.reads, back):
        assert (a.query_name, a.reference_name, a.reference_start, a.mapping_quality) == \
               (b.query_name, b.reference_name, b.reference_start, b.mapping_quality)
        assert a.cigartuples == b.cigartuples and a.query_sequence == b.query_sequence
        assert list(a.query_qualities) == list(b.query_qualities)
        assert a.tags["MD"] == b.tags["MD"]


def test_fasta_round_trip(tmp_path):
    g = {"c1": "ACGTNacgt" * 20, "c2": "GGG"}
    path = os.path.join(tmp_path, "g.fa")
    bamio.write_fasta(path, g)
    assert bamio.read_fasta(path) == g

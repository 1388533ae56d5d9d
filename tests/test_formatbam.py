"""formatbam (reference formatbam.py:61-94 + samtools.py:9-38): secondary / supplementary alignments
out, mate tags, coordinate sort, index -- on a BAM as an aligner writes it (grouped by query name).  And the packer's own
FLAG filter (`exclude_flags`), which does step 1 while it reads a file that never went through formatbam."""
import copy
import gzip
import os
import struct

import numpy as np
import pytest
from click.testing import CliRunner

from mustache_amd import bamio, formatbam, lib
from mustache_amd.main import cli
from mustache_amd.main import formatbam as formatbam_cmd
from mustache_amd.pack import ALL_FIELDS, pack_bam, pack_reads
from mustache_amd.records import AlignedRead
from tests import util


def _aligner_output(case_name, n_extra=12, seed=1):
    """Reads of a golden case as an aligner writes them: grouped by name (mates next to each other, a read without a mapped
    mate gets an unmapped one), with supplementary / secondary copies behind some of them.  Returns (contigs, records in
    file order, the clean primary records)."""
    case = util.load_json(case_name)
    rs = util.case_readset(case)
    rng = np.random.default_rng(seed)
    by_name = {}
    for r in rs.reads:
        by_name.setdefault(r.query_name, []).append(r)
    names = list(by_name)
    rng.shuffle(names)
    out, primaries = [], []
    extra_at = set(rng.choice(len(names), size=min(n_extra, len(names)), replace=False).tolist())
    for k, nm in enumerate(names):
        reads = by_name[nm][:2]
        for i, r in enumerate(reads):
            r.flag = (r.flag & ~0x900) | 1 | (0x40 if i == 0 else 0x80)
        if len(reads) == 1:   # the mate did not map
            m = AlignedRead(nm, None, -1, 0, None, "ACGTACGTAC", [30] * 10, flag=1 | 4 | 0x80, reference_id=-1)
            reads = [reads[0], m]
        for r in reads:
            out.append(r)
            if r.cigartuples is not None:
                primaries.append(r)
            if k in extra_at and r.cigartuples is not None:
                for fl in (0x800, 0x100):   # the same alignment again, flagged supplementary / secondary
                    x = copy.copy(r)
                    x.tags = dict(r.tags)
                    x.flag = r.flag | fl
                    out.append(x)
    return rs.contigs, out, primaries


def _write(path, contigs, reads, text=None):
    with bamio.BamWriter(path, contigs, text) as w:
        for r in reads:
            w.write(r)


def test_formatbam_paired(tmp_path):
    contigs, recs, primaries = _aligner_output("sim_a")
    raw, out = os.path.join(tmp_path, "raw.bam"), os.path.join(tmp_path, "out.bam")
    _write(raw, contigs, recs, text="@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "@PG\tID:aligner\n")
    assert any(r.flag & 0x900 for r in recs) and any(r.flag & 4 for r in recs)
    assert formatbam._formatbam(raw, out, False, False, verbose=False) == out
    with bamio.BamReader(out) as bam:
        assert "SO:coordinate" in bam.text.split("\n")[0] and "@PG\tID:aligner" in bam.text
        got = list(bam)
    assert len(got) == len(primaries)                                   # 0x900 and unmapped records are gone
    assert not any(r.flag & (0x900 | 4) for r in got)
    keys = [(r.reference_id, r.reference_start) for r in got]
    assert keys == sorted(keys)
    by_name = {}
    for r in primaries:
        by_name.setdefault(r.query_name, []).append(r)
    unmapped_mate = {r.query_name: r for r in recs if r.flag & 4}
    for r in got:                                                       # MT / MQ = the mate's sequence / quality string
        mates = [m for m in by_name[r.query_name] if (m.flag & 0xC0) != (r.flag & 0xC0)]
        m = mates[0] if mates else unmapped_mate[r.query_name]
        assert r.tags["MT"] == m.query_sequence
        assert r.tags["MQ"] == "".join(chr(int(q) + 33) for q in m.query_qualities)
        assert r.tags["MD"] == [p for p in by_name[r.query_name] if (p.flag & 0xC0) == (r.flag & 0xC0)][0].tags["MD"]
    # what findflanks packs from the formatted file = the clean primaries in samtools' sort order (position, then strand;
    # ties in file order)
    want = pack_reads(contigs, sorted(primaries, key=lambda r: (r.reference_id, r.reference_start, (r.flag >> 4) & 1)), pool="clipped")
    have = pack_bam(out, pool="clipped")
    for k, _ in ALL_FIELDS:
        assert np.array_equal(have.a[k], want.a[k]), k


def _sam_text(contigs, reads, header):
    """The reads as an aligner's SAM text (1-based POS, '*' for absent fields)."""
    from mustache_amd.records import cigar_to_string
    lines = [header.rstrip("\n")]
    for r in reads:
        mapped = r.cigartuples is not None
        tags = []
        for k, v in r.tags.items():
            tags.append("%s:Z:%s" % (k, v) if isinstance(v, str) else "%s:i:%d" % (k, int(v)))
        lines.append("\t".join([r.query_name, str(r.flag), contigs[r.reference_id][0] if mapped else "*",
                                str(r.reference_start + 1 if mapped else 0), str(r.mapping_quality if mapped else 0),
                                cigar_to_string(r.cigartuples) if mapped else "*", "*", "0", "0", r.query_sequence,
                                "".join(chr(int(q) + 33) for q in r.query_qualities)] + tags))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("gz", [False, True])
def test_formatbam_takes_sam_text(tmp_path, gz):
    """`samtools view -b -h -F 0x900 {in_sam}` (samtools.py:10) reads SAM text as well as BAM: the same aligner output as
    text gives the same formatted records (every field and tag) as it does as BAM."""
    contigs, recs, primaries = _aligner_output("sim_b", seed=3)
    header = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "@PG\tID:aligner\n"
    raw, sam = os.path.join(tmp_path, "raw.bam"), os.path.join(tmp_path, "raw.sam" + (".gz" if gz else ""))
    _write(raw, contigs, recs, text=header)
    text = _sam_text(contigs, recs, header).encode()
    with (gzip.open(sam, "wb") if gz else open(sam, "wb")) as fh:
        fh.write(text)
    out_b, out_s = os.path.join(tmp_path, "b.bam"), os.path.join(tmp_path, "s.bam")
    formatbam._formatbam(raw, out_b, False, False, verbose=False)
    formatbam._formatbam(sam, out_s, False, False, verbose=False)
    with bamio.BamReader(out_b) as b, bamio.BamReader(out_s) as s_:
        assert b.text == s_.text and b.contigs == s_.contigs
        rb, rs_ = list(b), list(s_)
    assert len(rb) == len(rs_) == len(primaries)
    for x, y in zip(rb, rs_):
        assert (x.query_name, x.flag, x.reference_id, x.reference_start, x.mapping_quality, x.cigartuples, x.query_sequence,
                list(x.query_qualities), x.tags) == \
               (y.query_name, y.flag, y.reference_id, y.reference_start, y.mapping_quality, y.cigartuples, y.query_sequence,
                list(y.query_qualities), y.tags)
    have, want = pack_bam(out_s, pool="clipped"), pack_bam(out_b, pool="clipped")
    for k, _ in ALL_FIELDS:
        assert np.array_equal(have.a[k], want.a[k]), k
    assert open(out_s + ".bai", "rb").read()[:4] == b"BAI\1"


def test_sam_text_fields_and_tag_types():
    rec = formatbam.sam_line_to_record("r1\t99\tc2\t101\t37\t5S10M2D3M\t=\t301\t350\tACGTNACGTACGTACGTA\t" + "I" * 18 +
                                       "\tNM:i:2\tXS:i:-5\tXB:i:70000\tMD:Z:10^AC3\tXT:A:U\tXF:f:1.5\tXA:B:s,1,-2,3", {"c1": 0, "c2": 1})
    ref_id, pos, l_name, mapq, bin_, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
    assert (ref_id, pos, l_name, mapq, n_cig, flag, l_seq, nref, npos, tlen) == (1, 100, 3, 37, 4, 99, 18, 1, 300, 350)
    assert bin_ == bamio._reg2bin(100, 115)
    tags = bamio._parse_tags(rec, 32 + 3 + 16 + 9 + 18)
    assert tags["NM"] == 2 and tags["XS"] == -5 and tags["XB"] == 70000 and tags["MD"] == "10^AC3" and tags["XT"] == "U"
    assert rec[32 + 3 + 16 + 9 + 18 + 2:32 + 3 + 16 + 9 + 18 + 3] == b"C"          # smallest integer type, as htslib stores it
    with pytest.raises(ValueError, match="@SQ"):
        formatbam.sam_line_to_record("r1\t0\tnope\t1\t0\t3M\t*\t0\t0\tACG\tIII", {"c1": 0})
    with pytest.raises(ValueError, match="11 are mandatory"):
        formatbam.sam_line_to_record("r1\t0\tc1", {"c1": 0})


def test_formatbam_single_end_and_errors(tmp_path):
    contigs, recs, primaries = _aligner_output("sim_a")
    raw, out = os.path.join(tmp_path, "raw.bam"), os.path.join(tmp_path, "out.bam")
    _write(raw, contigs, recs)
    formatbam._formatbam(raw, out, True, False, verbose=False)          # --single-end: no pairing, unmapped reads stay (sorted last)
    with bamio.BamReader(out) as bam:
        got = list(bam)
    n_unmapped = sum(1 for r in recs if r.flag & 4)
    assert len(got) == len(primaries) + n_unmapped
    assert all(r.flag & 4 for r in got[len(primaries):]) and not any("MT" in r.tags for r in got)
    # mismatched mates: the reference's fatal error (message on stdout, sys.exit())
    bad = [r for r in recs if not r.flag & 0x900][1:]
    _write(raw, contigs, bad)
    with pytest.raises(SystemExit):
        formatbam._formatbam(raw, out, False, False, verbose=False)


def _bgzf_record_at(path, voff):
    """The raw record that starts at a virtual file offset."""
    with open(path, "rb") as fh:
        fh.seek(voff >> 16)
        data = b""
        while len(data) < (voff & 0xFFFF) + 4 or len(data) < (voff & 0xFFFF) + 4 + struct.unpack_from("<i", data, voff & 0xFFFF)[0]:
            head = fh.read(18)
            bsize = struct.unpack_from("<H", head, 16)[0] + 1
            body = fh.read(bsize - 18)
            import zlib
            data += zlib.decompress(body[:-8], -15)
    o = voff & 0xFFFF
    size, = struct.unpack_from("<i", data, o)
    return data[o + 4:o + 4 + size]


def test_index_points_at_the_records_of_each_bin(tmp_path):
    contigs, recs, primaries = _aligner_output("sim_b", n_extra=30)
    raw, out = os.path.join(tmp_path, "raw.bam"), os.path.join(tmp_path, "out.bam")
    _write(raw, contigs, recs)
    formatbam._formatbam(raw, out, False, False, verbose=False)
    bai = open(out + ".bai", "rb").read()
    assert bai[:4] == b"BAI\1"
    n_ref, = struct.unpack_from("<i", bai, 4)
    assert n_ref == len(contigs)
    o, n_chunks, seen, n_meta = 8, 0, 0, 0
    names = [n for n, _ in contigs]
    placed = [0] * n_ref                                                    # records per reference (all mapped here)
    for r in bamio.BamReader(out):
        placed[names.index(r.reference_name)] += 1
    for c in range(n_ref):
        n_bin, = struct.unpack_from("<i", bai, o); o += 4
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, o); o += 8
            if b == 37450:                                                  # metadata pseudo-bin: record span, mapped / unmapped counts
                assert n_chunk == 2
                ref_beg, ref_end, n_mapped, n_unmapped = struct.unpack_from("<QQQQ", bai, o); o += 32
                assert ref_beg < ref_end and n_mapped == placed[c] > 0 and n_unmapped == 0
                assert struct.unpack_from("<i", _bgzf_record_at(out, ref_beg), 0)[0] == c
                n_meta += 1
                continue
            for _ in range(n_chunk):
                beg, end = struct.unpack_from("<QQ", bai, o); o += 16
                assert beg < end
                rec = _bgzf_record_at(out, beg)
                ref_id, pos, l_name, mapq, rbin, n_cig = struct.unpack_from("<iiBBHH", rec, 0)
                assert ref_id == c and rbin == b                          # the chunk starts at a record of this bin
                n_chunks += 1
        n_intv, = struct.unpack_from("<i", bai, o); o += 4
        offs = struct.unpack_from("<%dQ" % n_intv, bai, o); o += 8 * n_intv
        assert list(offs) == sorted(offs)                                 # the file is sorted: window offsets ascend
        for w, v in enumerate(offs):
            if v:
                rec = _bgzf_record_at(out, v)
                ref_id, pos = struct.unpack_from("<ii", rec, 0)
                assert ref_id == c
                seen += 1
    n_no_coor, = struct.unpack_from("<Q", bai, o)
    assert n_no_coor == 0 and o + 8 == len(bai) and n_chunks > 0 and seen > 0 and n_meta == sum(1 for n in placed if n)


def test_packers_skip_excluded_flags(tmp_path):
    """exclude_flags=0x900 while packing = `samtools view -F 0x900` first; native and Python packers agree."""
    contigs, recs, primaries = _aligner_output("sim_c")
    mapped = sorted([r for r in recs if r.cigartuples is not None], key=lambda r: (r.reference_id, r.reference_start))
    clean = [r for r in mapped if not r.flag & 0x900]
    path = os.path.join(tmp_path, "with_supp.bam")
    _write(path, contigs, mapped)
    want = pack_reads(contigs, clean, pool="clipped")
    for have in (pack_bam(path, pool="clipped", exclude_flags=0x900, threads=3, chunk_bytes=1 << 16),
                 pack_reads(contigs, mapped, pool="clipped", exclude_flags=0x900)):
        for k, _ in ALL_FIELDS:
            assert np.array_equal(have.a[k], want.a[k]), k
    assert pack_bam(path, pool="clipped").n_reads == len(mapped) > len(clean)


def test_formatbam_command_flags():
    opts = {p.name: p for p in formatbam_cmd.params}
    assert set(opts) == {"in_sam", "out_bam", "single_end", "keep_tmp_files"}           # reference main.py:262-269
    assert opts["single_end"].is_flag and opts["keep_tmp_files"].is_flag
    assert "--single-end" in opts["single_end"].opts and "--keep-tmp-files" in opts["keep_tmp_files"].opts


@pytest.mark.gpu
def test_supplementary_alignments_abort_findflanks_unless_filtered(tmp_path, hip):
    """A chimeric read: three alignments of one name clipped at one site.  The reference stops (findflanks.py:971-978) and so
    does the library; after formatbam, or with --exclude-flags 0x900, the run gives the golden table."""
    from mustache_amd.findflanks import _findflanks
    case = util.load_json("sim_a")
    rs = util.case_readset(case)
    clipped = []
    for line in case["findflanks_tsv"].strip().split("\n")[1:]:
        f = line.split("\t")
        contig, pos, orient = f[2], int(f[3]), f[4]
        for r in rs.reads:
            if r.reference_name != contig or r.mapping_quality < 20:
                continue
            if orient == "5p" and r.cigartuples[-1][0] == 4 and r.cigartuples[-1][1] >= 8 and r.reference_end == pos:
                clipped.append(r)
            if orient == "3p" and r.cigartuples[0][0] == 4 and r.cigartuples[0][1] >= 8 and r.reference_start - 1 == pos:
                clipped.append(r)
        if clipped:
            break
    assert clipped
    target = clipped[0]
    reads = []
    for r in rs.reads:
        reads.append(r)
        if r is target:
            for _ in range(2):
                x = copy.copy(r)
                x.tags = dict(r.tags)
                x.flag = r.flag | 0x800
                reads.append(x)
    bam = os.path.join(tmp_path, "chimeric.bam")
    _write(bam, rs.contigs, reads)
    p = {k: v for k, v in lib.DEFAULT_PARAMS.items() if k in ("min_softclip_length", "min_softclip_count", "min_alignment_quality",
                                                             "min_alignment_inner_length", "min_distance_to_mate", "min_softclip_ratio",
                                                             "max_indel_ratio", "large_insertion_cutoff", "min_count_consensus")}
    p.update(case["findflanks_params"])
    args = (p["min_softclip_length"], p["min_softclip_count"], p["min_alignment_quality"], p["min_alignment_inner_length"],
            p["min_distance_to_mate"], p["min_softclip_ratio"], p["max_indel_ratio"], p["large_insertion_cutoff"], p["min_count_consensus"], "sample1")
    with pytest.raises(SystemExit):                      # the reference prints the message and calls sys.exit() (findflanks.py:976-978)
        _findflanks(bam, *args, output_file=None, verbose=False)
    from mustache_amd.findflanks import device_reads
    from mustache_amd.pack import pack_bam
    with pytest.raises(lib.MfkError, match="READ NUMBER ERROR"):   # the C-ABI itself reports error 3
        hip.run_findflanks(device_reads(hip, pack_bam(bam, pool="clipped"), "cuda:0"), lib.make_params(**p))
    out = os.path.join(tmp_path, "ff.tsv")
    _findflanks(bam, *args, output_file=out, verbose=False, exclude_flags=0x900)
    assert open(out).read() == case["findflanks_tsv"]
    r = CliRunner().invoke(cli, ["findflanks", bam, "-o", out, "-id", "sample1", "--exclude-flags", "0x900"] +
                           [x for k, v in p.items() for x in ("--" + k, str(v))], catch_exceptions=False)
    assert r.exit_code == 0 and open(out).read() == case["findflanks_tsv"]

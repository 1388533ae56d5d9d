"""libmfkio.so (include/mfk_bamio.h): the native BGZF/BAM decoder + packer against the Python packer, array for array."""
import ctypes as C
import os
import struct
import re

import numpy as np
import pytest

from mustache_amd import bamio, pack
from mustache_amd.records import AlignedRead, ReadSet
from mustache_amd.simulate import simulate_readset
from tests import util

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mfk_bamio.h")


def same(p, q):
    assert p.contig_names == q.contig_names and p.max_span == q.max_span and p.n_reads == q.n_reads
    for k, t in pack.ALL_FIELDS:
        assert p.a[k].dtype == q.a[k].dtype == np.dtype(t), k
        assert np.array_equal(p.a[k], q.a[k]), (k, p.a[k][:10], q.a[k][:10])


def python_pack(path, pool, **kw):
    with bamio.BamReader(path) as bam:
        return pack.pack_reads(bam.contigs, bam, pool=pool, **kw)


def test_library_exports_every_declared_symbol():
    so = pack.io_lib()
    declared = re.findall(r"\b(mfkio_[a-z_]+)\s*\(", open(HEADER).read())
    assert len(set(declared)) >= 5
    for name in set(declared):
        assert hasattr(so, name), name
    assert so.mfkio_version() == 1


@pytest.mark.parametrize("case", util.CASES)
@pytest.mark.parametrize("pool", ["clipped", "all", "none"])
def test_golden_cases_pack_identically(tmp_path, case, pool):
    rs = util.case_readset(util.load_json(case))
    path = os.path.join(tmp_path, "x.bam")
    bamio.write_bam(path, rs)
    want = python_pack(path, pool)
    same(pack.pack_bam(path, pool=pool, threads=3), want)
    same(pack.pack_reads(rs.contigs, rs, pool=pool), want)        # the BAM round trip itself changes nothing


@pytest.mark.parametrize("threads,chunk", [(1, 0), (4, 1), (7, 70000), (8, 300000)])
def test_threads_and_straddling_passes(tmp_path, threads, chunk):
    """Tiny passes (one BGZF block each) make most records straddle two passes and the header span several."""
    genome, rs, truth = simulate_readset(seed=11, contig_lens=(9000, 2500, 700), n_insertions=6, coverage=50)
    path = os.path.join(tmp_path, "s.bam")
    bamio.write_bam(path, rs)
    want = python_pack(path, "clipped")
    got = pack.pack_bam(path, pool="clipped", threads=threads, chunk_bytes=chunk)
    same(got, want)
    st = got.io_stats
    assert st["n_records"] == len(rs.reads) and st["n_threads"] == threads
    assert st["uncompressed_bytes"] > st["compressed_bytes"] > 0


def test_zlib_and_libdeflate_agree(tmp_path):
    """Run the decoder in a child process with libdeflate switched off and compare with this process's result."""
    import subprocess
    import sys
    genome, rs, truth = simulate_readset(seed=12, contig_lens=(5000,), n_insertions=2, coverage=30)
    path = os.path.join(tmp_path, "z.bam")
    bamio.write_bam(path, rs)
    got = pack.pack_bam(path, pool="all")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from mustache_amd import pack; "
            "p = pack.pack_bam(%r, pool='all'); assert p.io_stats['inflate_backend'] == 0; "
            "np.savez(%r, **p.a)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path,
                                     os.path.join(tmp_path, "z.npz")))
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, MFK_NO_LIBDEFLATE="1"))
    z = np.load(os.path.join(tmp_path, "z.npz"))
    for k, _ in pack.ALL_FIELDS:
        assert np.array_equal(z[k], got.a[k]), k


def test_region_mode_matches_the_pairflanks_host_filter(tmp_path):
    """pool='regions' == the generator filter + in_window pool callable of mustache_amd/pairflanks.py::_pairflanks."""
    genome, rs, truth = simulate_readset(seed=13, contig_lens=(8000, 3000), n_insertions=5, coverage=40)
    path = os.path.join(tmp_path, "r.bam")
    bamio.write_bam(path, rs)
    rng = np.random.default_rng(5)
    regions = []
    for _ in range(12):
        cid = int(rng.integers(0, 2))
        start = int(rng.integers(0, rs.contigs[cid][1] - 60))
        end = start + int(rng.integers(1, 22))
        center = round((end + start) / 2)
        es, ee = max(center - 25, 0), min(center + 25, rs.contigs[cid][1])
        regions.append((cid, min(es, start - 1), max(ee, end + 2), es, ee))
    regions.append((1, 500, 400, 450, 420))                       # inverted intervals keep the Python meaning
    by = {}
    for cid, lo, hi, es, ee in regions:
        by.setdefault(cid, []).append((lo, hi, es, ee))
    index = {n: i for i, (n, _) in enumerate(rs.contigs)}

    def near(r):
        return any(r.reference_start < hi and r.reference_end > lo for lo, hi, _, _ in by.get(index[r.reference_name], ()))

    def in_window(r, cid):
        return any(r.reference_start < ee and r.reference_end > es for _, _, es, ee in by.get(cid, ()))
    with bamio.BamReader(path) as bam:
        want = pack.pack_reads(bam.contigs, (r for r in bam if r.cigartuples is not None and near(r)), pool=in_window)
    got = pack.pack_bam(path, pool="regions", regions=regions, threads=4, chunk_bytes=100000)
    assert 0 < want.n_reads < len(rs.reads)
    same(got, want)


def _write(tmp_path, contigs, reads, name="e.bam"):
    path = os.path.join(tmp_path, name)
    with bamio.BamWriter(path, contigs) as w:
        for r in reads:
            w.write(r)
    return path


def test_edge_records(tmp_path):
    """Unmapped reads, hard clips, H+S, insertions, deletions, padding, odd lengths, missing qualities, long names,
    three reads of one name, reads on the last contig only, an empty contig in between."""
    q = lambda n: [(7 * k) % 41 for k in range(n)]
    reads = [
        AlignedRead("u0", "c1", 3, 0, None, "ACGT", q(4), flag=4),
        AlignedRead("a" * 200, "c1", 5, 60, [(5, 3), (4, 2), (0, 9)], "ACGTTGCAACG", q(11), md="9"),
        AlignedRead("b", "c1", 5, 60, [(4, 3), (0, 4), (1, 2), (0, 5), (2, 3), (0, 2), (5, 1)], "ACGTTGCAACGTTGCA", q(16), md="9^ACG2"),
        AlignedRead("b", "c1", 6, 13, [(0, 4), (6, 1), (3, 10), (7, 3), (8, 1)], "NACGTRYK", None, md="7A0"),
        AlignedRead("c", "c1", 7, 60, [(0, 7)], "ACGTACG", q(7), md="0A5C0"),
        AlignedRead("b", "c1", 9, 60, [(0, 5), (4, 4)], "ACGTACGTA", q(9), md="5"),
        AlignedRead("d", "c3", 0, 60, [(0, 3)], "ACG", q(3), md="3"),
    ]
    contigs = [("c1", 100), ("c2", 50), ("c3", 10)]
    for r in reads:
        r.reference_id = {"c1": 0, "c2": 1, "c3": 2}[r.reference_name]
    path = _write(tmp_path, contigs, reads)
    for pool in ("all", "clipped", "none"):
        want = python_pack(path, pool)
        same(pack.pack_bam(path, pool=pool, threads=2), want)
    p = pack.pack_bam(path, pool="all")
    assert p.n_reads == 6 and list(p.link) == [-1, -1, 1, -1, 2, -1]
    assert list(p.contig_read_start) == [0, 5, 5, 6]


def test_empty_and_header_only(tmp_path):
    path = _write(tmp_path, [("c1", 100)], [])
    want = python_pack(path, "clipped")
    got = pack.pack_bam(path)
    same(got, want)
    assert got.n_reads == 0 and got.io_stats["n_records"] == 0


def test_rejections_match_the_python_packer(tmp_path):
    ok = dict(query_sequence="ACGTACGTAC", query_qualities=[30] * 10)
    bad = {
        "unsorted": [AlignedRead("a", "c1", 50, 60, [(0, 10)], md="10", **ok),
                     AlignedRead("b", "c1", 10, 60, [(0, 10)], md="10", **ok)],
        "no_md": [AlignedRead("a", "c1", 5, 60, [(0, 10)], **ok)],
        "starts_with_deletion": [AlignedRead("a", "c1", 5, 60, [(2, 2), (0, 10)], md="^AC10", **ok)],
        "inner_clip": [AlignedRead("a", "c1", 5, 60, [(0, 4), (4, 2), (0, 4)], md="8", **ok)],
    }
    for name, reads in bad.items():
        for r in reads:
            r.reference_id = 0
        path = _write(tmp_path, [("c1", 100)], reads, name + ".bam")
        with pytest.raises(pack.PackError):
            python_pack(path, "clipped")
        with pytest.raises(pack.PackError):
            pack.pack_bam(path)
    # both S ends: no MD needed
    r = AlignedRead("a", "c1", 5, 60, [(4, 2), (0, 6), (4, 2)], **ok)
    r.reference_id = 0
    path = _write(tmp_path, [("c1", 100)], [r], "s_only.bam")
    same(pack.pack_bam(path), python_pack(path, "clipped"))


def test_corrupt_files_are_clear_errors(tmp_path):
    genome, rs, truth = simulate_readset(seed=14, contig_lens=(3000,), n_insertions=1, coverage=20)
    path = os.path.join(tmp_path, "c.bam")
    bamio.write_bam(path, rs)
    data = bytearray(open(path, "rb").read())
    trunc = os.path.join(tmp_path, "trunc.bam")
    open(trunc, "wb").write(data[:len(data) // 2])
    with pytest.raises(ValueError):
        pack.pack_bam(trunc)
    flip = os.path.join(tmp_path, "flip.bam")
    data[len(data) // 2] ^= 0x55
    open(flip, "wb").write(data)
    with pytest.raises(ValueError):
        pack.pack_bam(flip)
    text = os.path.join(tmp_path, "t.bam")
    open(text, "wb").write(b"this is not a BAM file at all, not even gzip\n")
    with pytest.raises(ValueError):
        pack.pack_bam(text)
    with pytest.raises(OSError):
        pack.pack_bam(os.path.join(tmp_path, "missing.bam"))


@pytest.mark.parametrize("case", util.CASES)
def test_pairflanks_regions_of_the_golden_cases(tmp_path, case):
    """The regions _pairflanks derives from a golden findflanks table select and pool the same reads in both decoders."""
    import io

    import pandas as pd
    from mustache_amd import lib, pairflanks as pfh
    c = util.load_json(case)
    rs = util.case_readset(c)
    path = os.path.join(tmp_path, "p.bam")
    bamio.write_bam(path, rs)
    flanks = pd.read_csv(io.StringIO(c["findflanks_tsv"]), sep="\t")
    names = [n for n, _ in rs.contigs]
    tab, _ = pfh.flanks_to_table(flanks, names)
    g, goff = pfh.genome_arrays(c["genome"], names)
    regions = pfh.candidate_regions(tab, lib.make_params(**c["pairflanks_params"]).max_direct_repeat_length, np.diff(goff))
    with bamio.BamReader(path) as bam:
        want = pfh._pack_regions_python(bam, names, regions)
    got = pack.pack_bam(path, pool="regions", regions=regions, threads=3)
    same(got, want)
    if len(regions):
        assert 0 < got.n_reads < len(rs.reads)


def test_replicated_bam_repeats_the_base_tables(tmp_path, oracle):
    """bigbam: k copies of a read set on k contig sets -> the base findflanks table k times (oracle on native packs)."""
    from mustache_amd import bigbam, lib
    from mustache_amd.findflanks import flank_frame
    genome, rs, truth = simulate_readset(seed=21, contig_lens=(5000, 1500), n_insertions=4, coverage=40)
    path = os.path.join(tmp_path, "big.bam")
    contigs, n = bigbam.write_replicated_bam(path, rs, copies=5, procs=2)
    assert n == 5 * len(rs.reads)
    got = pack.pack_bam(path, pool="clipped", threads=4, chunk_bytes=1 << 20)
    same(got, python_pack(path, "clipped"))
    assert got.n_reads == 5 * len(rs.reads) and int((got.link >= 0).sum()) == 5 * int(
        (pack.pack_reads(rs.contigs, rs, pool="none").link >= 0).sum())
    base = pack.pack_reads(rs.contigs, rs, pool="clipped")
    params = lib.make_params()
    want = bigbam.replicate_table(flank_frame(oracle.run_findflanks(oracle.reads(base), params), base.contig_names), 5)
    have = flank_frame(oracle.run_findflanks(oracle.reads(got), params), got.contig_names)
    assert len(want) >= 5 * 2 * len(truth) and want.to_csv(sep="\t") == have.to_csv(sep="\t")


def test_wrong_share_guess_is_caught(tmp_path):
    """A record whose 400 KB array tag is itself a run of well-formed alignment records: the share that starts inside it
    guesses a record boundary that is not one; the verification notices and the pass is walked from the front."""
    import struct
    genome, rs, truth = simulate_readset(seed=15, contig_lens=(4000,), n_insertions=2, coverage=30)
    fake = b"".join(bamio.encode_record(r, 0) for r in rs.reads[:40])
    fake = fake * (400000 // len(fake) + 1)
    mid = rs.reads[len(rs.reads) // 2]
    big = AlignedRead("giant", "c1", mid.reference_start, 60, [(0, 20)], "ACGTACGTACGTACGTACGT", [30] * 20, md="20")
    big.reference_id = 0
    rec = bytearray(bamio.encode_record(big, 0))
    rec += b"ZBBC" + struct.pack("<i", len(fake)) + fake
    rec[0:4] = struct.pack("<i", len(rec) - 4)
    path = os.path.join(tmp_path, "g.bam")
    with bamio.BamWriter(path, rs.contigs) as w:
        for k, r in enumerate(rs.reads):
            if k == len(rs.reads) // 2:
                w._put(bytes(rec))
            w.write(r)
    want = python_pack(path, "clipped")
    assert want.n_reads == len(rs.reads) + 1
    for threads in (2, 4, 8):
        got = pack.pack_bam(path, pool="clipped", threads=threads)
        same(got, want)
        assert got.io_stats["n_records"] == len(rs.reads) + 1 and got.io_stats["n_redone_passes"] == 1


# ---- record index (mfkio_pack_bam_index / mfkio_pack_bam_regions_indexed) ------------------------------------------------
def _regions_around(rs, truth, names, extra=()):
    idx = {n: i for i, n in enumerate(names)}
    regs = []
    for t in truth:
        cid, start, end = idx[t["contig"]], t["pos_3p"], t["pos_5p"]
        center = round((end + start) / 2)
        es, ee = max(center - 25, 0), center + 25
        regs.append((cid, min(es, start - 1), max(ee, end + 2), es, ee))
    return regs + list(extra)


@pytest.mark.parametrize("threads,chunk", [(1, 0), (4, 1), (5, 70000)])
def test_indexed_region_pass_equals_full_region_pass(tmp_path, threads, chunk):
    """The region pass through the record index that a full pass left behind packs the same arrays as the full region pass:
    regions at insertion sites, at both contig ends, on a contig without reads nearby, overlapping regions, none at all."""
    genome, rs, truth = simulate_readset(seed=21, contig_lens=(30000, 12000, 900), n_insertions=5, coverage=60)
    path, idx = os.path.join(tmp_path, "s.bam"), os.path.join(tmp_path, "s.mfki")
    bamio.write_bam(path, rs)
    names = [n for n, _ in rs.contigs]
    full = pack.pack_bam(path, pool="clipped", threads=threads, chunk_bytes=chunk, index_out=idx)
    same(full, pack.pack_bam(path, pool="clipped", threads=threads, chunk_bytes=chunk))        # writing the index changes nothing
    assert os.path.exists(idx) and not os.path.exists(idx + ".tmp")
    extra = [(0, -5, 60, 0, 40), (0, 29900, 30050, 29950, 30000), (1, 0, 12000, 100, 150), (2, 300, 420, 320, 370), (0, 15000, 15010, 15000, 15010)]
    for regs in (_regions_around(rs, truth, names), _regions_around(rs, truth, names, extra), extra[:1], extra[2:3], []):
        want = pack.pack_bam(path, pool="regions", regions=regs, threads=3)
        got = pack.pack_bam(path, pool="regions", regions=regs, threads=threads, index=idx)
        assert got.io_stats["indexed"] and not want.io_stats["indexed"]
        same(got, want)
        if regs and regs != extra[2:3]:
            assert got.io_stats["uncompressed_bytes"] < full.io_stats["uncompressed_bytes"]     # it inflated less than the file


def test_indexed_pass_inflates_a_small_part_of_a_deep_file(tmp_path):
    genome, rs, truth = simulate_readset(seed=22, contig_lens=(60000,), n_insertions=3, coverage=150)
    path, idx = os.path.join(tmp_path, "d.bam"), os.path.join(tmp_path, "d.mfki")
    bamio.write_bam(path, rs)
    full = pack.pack_bam(path, pool="clipped", index_out=idx)
    regs = _regions_around(rs, truth, [n for n, _ in rs.contigs])
    got = pack.pack_bam(path, pool="regions", regions=regs, index=idx)
    same(got, pack.pack_bam(path, pool="regions", regions=regs))
    assert got.io_stats["indexed"] and got.io_stats["uncompressed_bytes"] * 5 < full.io_stats["uncompressed_bytes"]


def test_stale_or_foreign_index_falls_back_to_the_full_pass(tmp_path):
    genome, rs, truth = simulate_readset(seed=23, contig_lens=(9000,), n_insertions=2, coverage=40)
    path, idx = os.path.join(tmp_path, "a.bam"), os.path.join(tmp_path, "a.mfki")
    bamio.write_bam(path, rs)
    pack.pack_bam(path, pool="clipped", index_out=idx)
    regs = _regions_around(rs, truth, [n for n, _ in rs.contigs])
    want = pack.pack_bam(path, pool="regions", regions=regs)
    # another FLAG filter than the index was made with
    got = pack.pack_bam(path, pool="regions", regions=regs, index=idx, exclude_flags=0x900)
    assert not got.io_stats["indexed"]
    same(got, pack.pack_bam(path, pool="regions", regions=regs, exclude_flags=0x900))
    # the BAM was rewritten after the index (other mtime, possibly other size)
    genome2, rs2, truth2 = simulate_readset(seed=24, contig_lens=(9000,), n_insertions=2, coverage=41)
    bamio.write_bam(path, rs2)
    os.utime(path, ns=(1, 1))
    got = pack.pack_bam(path, pool="regions", regions=regs, index=idx)
    assert not got.io_stats["indexed"]
    same(got, pack.pack_bam(path, pool="regions", regions=regs))
    # garbage, a truncated index, no index at all
    for blob in (b"not an index", open(idx, "rb").read()[:100], None):
        bad = os.path.join(tmp_path, "bad.mfki")
        if blob is None:
            os.remove(bad)
        else:
            open(bad, "wb").write(blob)
        got = pack.pack_bam(path, pool="regions", regions=regs, index=bad)
        assert not got.io_stats["indexed"]
    # an index of the right file whose offsets were damaged (first block not at 0, a mark beyond the stream): refused on load
    pack.pack_bam(path, pool="clipped", index_out=idx)
    good = bytearray(open(idx, "rb").read())
    n_marks, n_blocks = struct.unpack_from("<QQ", good, 8 + 32 + 8)
    assert pack.pack_bam(path, pool="regions", regions=regs, index=idx).io_stats["indexed"]
    for off, val in ((8 + 32 + 8 + 16 + 16 * n_marks + 8, 7),                          # uncompressed offset of block 0
                     (8 + 32 + 8 + 16 + 16 * (n_marks - 1), 1 << 40)):                    # stream offset of the last mark
        dmg = bytearray(good)
        struct.pack_into("<Q", dmg, off, val)
        open(bad, "wb").write(bytes(dmg))
        got = pack.pack_bam(path, pool="regions", regions=regs, index=bad)
        assert not got.io_stats["indexed"]
        same(got, pack.pack_bam(path, pool="regions", regions=regs))
    # inverted regions keep the Python meaning through the full pass
    got = pack.pack_bam(path, pool="regions", regions=regs + [(0, 500, 400, 450, 420)], index=idx)
    assert not got.io_stats["indexed"]
    del want


def test_index_path_that_cannot_be_written_is_not_an_error(tmp_path):
    genome, rs, truth = simulate_readset(seed=25, contig_lens=(5000,), n_insertions=1, coverage=20)
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    got = pack.pack_bam(path, pool="clipped", index_out=os.path.join(tmp_path, "no_such_dir", "a.mfki"))
    same(got, pack.pack_bam(path, pool="clipped"))

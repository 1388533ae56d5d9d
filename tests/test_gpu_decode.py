"""BAM decode on the device (mfk_decode_bam, mustache_amd/gpudecode.py): every array equal to the host packer's
(pack.pack_bam = libmfkio.so, itself checked against the Python statement pack.pack_reads in test_bamio_native.py), the same
rejections, the record index it leaves, and the findflanks table through it."""
import os
import struct

import numpy as np
import pytest

from mustache_amd import bamio, gpudecode, gpuinflate, lib, pack
from mustache_amd.pack import ALL_FIELDS, pack_bam
from mustache_amd.records import AlignedRead
from mustache_amd.simulate import simulate_readset
from tests import util

pytestmark = pytest.mark.gpu


def _same(dev, host):
    got = dev.to_host()
    assert got.n_reads == host.n_reads and got.max_span == host.max_span and got.contig_names == host.contig_names
    for k, _ in ALL_FIELDS:
        assert np.array_equal(got.a[k], host.a[k]), k


@pytest.mark.parametrize("name", util.CASES)
@pytest.mark.parametrize("pool", ["clipped", "all", "none"])
def test_golden_cases_equal_the_host_packer(hip, tmp_path, name, pool):
    rs = util.case_readset(util.load_json(name))
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    _same(gpudecode.decode_bam(hip, path, pool=pool), pack_bam(path, pool=pool))


@pytest.mark.parametrize("seed,lens,cov", [(3, (9000, 4000, 700), 60), (4, (60000,), 150), (5, (300, 20000, 300, 5000), 25)])
def test_simulated_sets_equal_the_host_packer(hip, tmp_path, seed, lens, cov):
    genome, rs, truth = simulate_readset(seed=seed, contig_lens=lens, n_insertions=3, coverage=cov, sub_rate=0.01, wildtype_frac=0.2)
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    for pool in ("clipped", "all"):
        for fl in (0, 0x900):
            _same(gpudecode.decode_bam(hip, path, pool=pool, exclude_flags=fl), pack_bam(path, pool=pool, exclude_flags=fl))


def _write(tmp_path, contigs, reads, name="e.bam"):
    path = os.path.join(tmp_path, name)
    index = {n: i for i, (n, _) in enumerate(contigs)}
    for r in reads:
        r.reference_id = index[r.reference_name]
    with bamio.BamWriter(path, contigs) as w:
        for r in reads:
            w.write(r)
    return path


def test_edge_records_equal_the_host_packer(hip, tmp_path):
    """Unmapped reads, hard clips, H+S, insertions, deletions, padding, =/X, odd lengths, missing qualities, long names,
    three reads of one name, an empty contig in between (the records of tests/test_bamio_native.py::test_edge_records)."""
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
    path = _write(tmp_path, [("c1", 100), ("c2", 50), ("c3", 10)], reads)
    for pool in ("all", "clipped", "none"):
        _same(gpudecode.decode_bam(hip, path, pool=pool), pack_bam(path, pool=pool))
    d = gpudecode.decode_bam(hip, path, pool="all").to_host()
    assert d.n_reads == 6 and list(d.link) == [-1, -1, 1, -1, 2, -1] and list(d.contig_read_start) == [0, 5, 5, 6]
    # a file with a header and no record
    path = _write(tmp_path, [("c1", 1000)], [], "none.bam")
    d = gpudecode.decode_bam(hip, path)
    assert d.n_reads == 0 and d.contig_read_start.tolist() == [0, 0] and d.io_stats["n_records"] == 0


def test_rejections_are_the_packers(hip, tmp_path):
    ok = dict(query_sequence="ACGTACGTAC", query_qualities=[30] * 10)
    bad = {
        "coordinate-sorted": [AlignedRead("a", "c1", 50, 60, [(0, 10)], md="10", **ok), AlignedRead("b", "c1", 10, 60, [(0, 10)], md="10", **ok)],
        "without MD tag": [AlignedRead("a", "c1", 5, 60, [(0, 10)], **ok)],
        "deletion": [AlignedRead("a", "c1", 5, 60, [(2, 2), (0, 10)], md="^AC10", **ok)],
        "clip operation inside": [AlignedRead("a", "c1", 5, 60, [(0, 4), (4, 2), (0, 4)], md="8", **ok)],
    }
    for k, (msg, reads) in enumerate(bad.items()):
        path = _write(tmp_path, [("c1", 100)], reads, "bad%d.bam" % k)
        with pytest.raises(pack.PackError, match=msg):
            gpudecode.decode_bam(hip, path)
        with pytest.raises(pack.PackError, match=msg):
            pack_bam(path, pool="clipped")
    r = AlignedRead("a", "c1", 5, 60, [(4, 2), (0, 6), (4, 2)], **ok)       # both ends S: no MD needed
    path = _write(tmp_path, [("c1", 100)], [r], "s_only.bam")
    _same(gpudecode.decode_bam(hip, path), pack_bam(path, pool="clipped"))
    path = os.path.join(tmp_path, "junk.bam")
    open(path, "wb").write(b"this is not a bam file, not even gzip")
    with pytest.raises(lib.MfkError):
        gpudecode.decode_bam(hip, path)


def test_large_file_many_blocks_and_its_record_index(hip, tmp_path):
    """Thousands of BGZF blocks (every block guesses its first record), records that straddle blocks, and the record index
    the device decode leaves: pairflanks' indexed region pass through it equals the full region pass."""
    from mustache_amd import bigbam
    genome, rs, truth = simulate_readset(seed=11, contig_lens=(30000, 8000), n_insertions=5, coverage=60)
    path, idx = os.path.join(tmp_path, "big.bam"), os.path.join(tmp_path, "big.mfki")
    contigs, n_rec = bigbam.write_replicated_bam(path, rs, 40, procs=4)
    host = pack_bam(path, pool="clipped")
    dev = gpudecode.decode_bam(hip, path, pool="clipped", index_out=idx)
    assert dev.io_stats["n_records"] == n_rec and dev.io_stats["uncompressed_bytes"] == host.io_stats["uncompressed_bytes"]
    _same(dev, host)
    assert open(idx, "rb").read(8) == b"MFKI0002"
    regs = [(c, 1000, 3000, 1500, 1550) for c in range(0, len(contigs), 7)]
    a = pack_bam(path, pool="regions", regions=regs, index=idx)
    b = pack_bam(path, pool="regions", regions=regs)
    assert a.io_stats["indexed"] and not b.io_stats["indexed"]
    for k, _ in ALL_FIELDS:
        assert np.array_equal(a.a[k], b.a[k]), k


@pytest.mark.parametrize("name", ["sim_a", "sim_c"])
def test_findflanks_through_the_device_decoder(hip, tmp_path, name):
    from mustache_amd.findflanks import _findflanks
    case = util.load_json(name)
    rs = util.case_readset(case)
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    d = dict(min_softclip_length=8, min_softclip_count=2, min_alignment_quality=20, min_alignment_inner_length=21, min_distance_to_mate=22,
             min_softclip_ratio=0.10, max_indel_ratio=0.03, large_insertion_cutoff=30, min_count_consensus=2)
    d.update(case["findflanks_params"])
    args = list(d.values())
    out_g, out_h = os.path.join(tmp_path, "g.tsv"), os.path.join(tmp_path, "h.tsv")
    _findflanks(path, *args, "s", out_g, verbose=False, decoder="gpu")
    _findflanks(path, *args, "s", out_h, verbose=False, decoder="native")
    assert open(out_g).read() == open(out_h).read() and len(open(out_g).read().splitlines()) > 1
    want = case["findflanks_tsv"].splitlines()      # the reference's own table (its sample column holds another name)
    assert [ln.split("\t")[1:] for ln in open(out_g).read().splitlines()] == [ln.split("\t")[1:] for ln in want]
    assert os.path.exists(out_g + ".mfki")


def test_checksum_of_every_block_is_checked(hip, tmp_path):
    """The host reader verifies the CRC-32 of every BGZF block; so does the device decode."""
    genome, rs, truth = simulate_readset(seed=41, contig_lens=(30000,), n_insertions=2, coverage=60)
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    data = bytearray(open(path, "rb").read())
    blocks, crcs, total = gpuinflate.bgzf_blocks(bytes(data))
    assert blocks.shape[0] > 3
    b = blocks[2]
    trailer = int(b["comp_off"]) + int(b["comp_len"])             # CRC32 | ISIZE behind the deflate stream
    data[trailer] ^= 0x01
    open(path, "wb").write(bytes(data))
    with pytest.raises(lib.MfkError, match="checksum"):
        gpudecode.decode_bam(hip, path)
    with pytest.raises(ValueError, match="checksum"):
        pack_bam(path, pool="clipped")
    assert gpudecode.decode_bam(hip, path, check_crc=False).n_reads > 0


def test_truncated_and_damaged_files_fail_like_the_host_reader(hip, tmp_path):
    """A file cut in the middle of a block, one cut between blocks (no EOF marker, a record cut off) and one with a damaged
    deflate stream: an error from both readers, never a table (the upload thread is joined before the error leaves)."""
    genome, rs, truth = simulate_readset(seed=42, contig_lens=(30000,), n_insertions=2, coverage=60)
    path = os.path.join(tmp_path, "a.bam")
    bamio.write_bam(path, rs)
    data = open(path, "rb").read()
    blocks, crcs, total = gpuinflate.bgzf_blocks(data)
    assert blocks.shape[0] > 4
    cut_mid = int(blocks[3]["comp_off"]) + int(blocks[3]["comp_len"]) // 2
    cut_between = int(blocks[3]["comp_off"]) - 18                 # (the 18-byte header of block 3 starts here)
    damaged = bytearray(data)
    damaged[int(blocks[2]["comp_off"]) + 7] ^= 0xFF
    for name, body in (("mid", data[:cut_mid]), ("between", data[:cut_between]), ("damaged", bytes(damaged))):
        p = os.path.join(tmp_path, name + ".bam")
        open(p, "wb").write(body)
        with pytest.raises((lib.MfkError, ValueError)):
            gpudecode.decode_bam(hip, p)
        with pytest.raises((lib.MfkError, ValueError)):
            pack_bam(p, pool="clipped")

"""The packer's per-read fields against what the reference's sctools.py computed (golden unit vectors)."""
import numpy as np
import pytest

from mustache_amd import pack
from mustache_amd.records import ReadSet, AlignedRead
from tests import util


def test_clip_fields_match_sctools():
    rows = util.load_json("unit_vectors")["sctools"]
    reads = [util.read_from_json(r["read"]) for r in rows]
    rs = ReadSet([("c1", 600)], reads)
    p = pack.pack_reads(rs.contigs, rs, pool="all")
    order = {id(r): k for k, r in enumerate(rs.reads)}
    for row, read in zip(rows, reads):
        k = order[id(read)]
        f = int(p.flags[k])
        lk, rk = f & 3, (f >> 2) & 3
        assert bool(lk) == row["is_left"] and bool(rk) == row["is_right"], row["read"]
        assert int(p.first_op_len[k]) == row["left_len"] and int(p.last_op_len[k]) == row["right_len"]
        seq, quals = p.pool_sequence(k)
        llen = int(p.first_op_len[k]) if lk == 1 else (1 if lk == 2 else 0)
        rlen = int(p.last_op_len[k]) if rk == 1 else (1 if rk == 2 else 0)
        assert seq[:llen] == row["left_seq"] and quals[:llen] == row["left_qual"]
        assert (seq[len(seq) - rlen:] if rlen else "") == row["right_seq"]
        assert (quals[len(seq) - rlen:] if rlen else []) == row["right_qual"]
        if lk:
            site = int(p.ref_start[k]) - 1 if lk == 1 else int(p.ref_start[k])
            assert ["c1", site] == row["left_site"]
        if rk:
            site = int(p.ref_end[k]) if rk == 1 else int(p.ref_end[k]) - 1
            assert ["c1", site] == row["right_site"]
        for minial, key in ((21, "inner21"), (45, "inner45")):
            ok = not (lk and rk) or (int(p.qlen[k]) - llen - rlen >= minial)
            assert ok == row[key]
        # blocks
        if f & pack.FLAG_COMPLEX:
            c = int(np.searchsorted(p.cx_read, k))
            o0, o1 = int(p.cx_off[c]), int(p.cx_off[c + 1])
            blocks = [[int(a), int(b)] for a, b in zip(p.blk_start[o0:o1], p.blk_end[o0:o1])]
        else:
            blocks = [[int(p.ref_start[k]), int(p.ref_end[k])]]
        assert blocks == [list(b) for b in row["blocks"]]


def test_links_and_order():
    case = util.load_json("quirks")
    rs = util.case_readset(case)
    p = pack.pack_reads(rs.contigs, rs, pool="clipped", keep_names=True)
    last = {}
    for k, name in enumerate(p.read_names):
        assert int(p.link[k]) == last.get(name, -1)
        last[name] = k
    assert np.all(np.diff(p.contig_id.astype(np.int64)) >= 0)
    assert pack.BYTES_PER_READ == 22


def test_unsorted_input_is_rejected():
    a = AlignedRead("a", "c1", 50, 60, [(0, 10)], "A" * 10, [30] * 10, md="10")
    b = AlignedRead("b", "c1", 10, 60, [(0, 10)], "A" * 10, [30] * 10, md="10")
    with pytest.raises(pack.PackError):
        pack.pack_reads([("c1", 100)], [a, b])


def test_slice_is_self_contained(oracle):
    from mustache_amd import lib
    case = util.load_json("sim_a")
    rs = util.case_readset(case)
    p = pack.pack_reads(rs.contigs, rs, pool="all")
    lo = int(p.contig_read_start[1])
    sub = p.slice_reads(lo, p.n_reads)
    res_full = oracle.run_findflanks(oracle.reads(p), lib.make_params())
    res_sub = oracle.run_findflanks(oracle.reads(sub), lib.make_params())
    full = res_full["flanks"][res_full["flanks"]["contig_id"] == 1]
    assert np.array_equal(full["pos"], res_sub["flanks"]["pos"])
    assert np.array_equal(full["consensus_softclip_count"], res_sub["flanks"]["consensus_softclip_count"])


def test_candidate_regions_equal_the_per_flank_loop():
    """pairflanks.candidate_regions (searchsorted over the 3p flanks) against the loop it replaces: for each 5p flank the 3p
    flanks of its contig in [pos5 - maxdr - 1, pos5) (pairflanks.py:155-157) and the 50-bp window around round() of the
    middle (pairflanks.py:294-305, half to even)."""
    import numpy as np
    from mustache_amd import lib, pairflanks as pfh
    rng = np.random.default_rng(4)
    for maxdr in (0, 3, 20):
        n = 400
        tab = np.zeros(n, dtype=lib.FLANK_DTYPE)
        tab["contig_id"] = rng.integers(0, 3, n)
        tab["pos"] = rng.integers(-1, 300, n)
        tab["orient"] = rng.choice([5, 3, 0], n)
        glen = np.array([300, 40, 10_000])
        want = set()
        five, three = tab[tab["orient"] == 5], tab[tab["orient"] == 3]
        for f5 in five:
            m = three[(three["contig_id"] == f5["contig_id"]) & (three["pos"] >= f5["pos"] - maxdr - 1) & (three["pos"] < f5["pos"])]
            for f3 in m:
                start, end = int(f3["pos"]), int(f5["pos"])
                center = round((end + start) / 2)
                es = max(int(center - 50 / 2), 0)
                ee = min(int(center + 50 / 2), int(glen[int(f5["contig_id"])]))
                want.add((int(f5["contig_id"]), min(es, start - 1), max(ee, end + 2), es, ee))
        got = pfh.candidate_regions(tab, maxdr, glen)
        assert got.dtype == np.int32 and got.shape[1] == 5
        assert set(map(tuple, got.tolist())) == want and (len(want) > 0 or maxdr == 0)


def test_clip_records_restate_the_columns():
    """PackedReads.clip_records (include/mfk.h mfk_clip_rec / clip_tile) against the per-read statement."""
    from mustache_amd.synth import CONFIGS, synth_packed
    from mustache_amd.pack import CLIP_TILE
    cfg = dict(CONFIGS["tiny"])
    cfg["contig_lens"] = (9000, 5000, 3000)
    packed, _, _ = synth_packed(**cfg, seed=5)
    a = packed.a
    rec, tile = packed.clip_records()
    want = [r for r in range(packed.n_reads) if a["flags"][r] & 15]
    assert rec["read"].tolist() == want and len(want) > 50
    assert tile.shape[0] == (packed.n_reads + CLIP_TILE - 1) // CLIP_TILE + 1
    for t in range(tile.shape[0]):
        assert tile[t] == sum(1 for r in want if r < t * CLIP_TILE)
    both = 0
    for k, r in enumerate(want):
        f = int(a["flags"][r])
        lk, rk = f & 3, (f >> 2) & 3
        inner = 255
        if lk and rk:
            both += 1
            inner = min(255, int(a["qlen"][r]) - (int(a["first_op_len"][r]) if lk == 1 else 1)
                        - (int(a["last_op_len"][r]) if rk == 1 else 1))
        assert (int(rec["ref_start"][k]), int(rec["ref_end"][k]), int(rec["contig_id"][k]), int(rec["mapq"][k]), int(rec["inner"][k])) == \
            (int(a["ref_start"][r]), int(a["ref_end"][r]), int(a["contig_id"][r]), int(a["mapq"][r]), inner)
    empty = packed.take_ranges([]) if hasattr(packed, "take_ranges") else None
    if empty is not None:
        r0, t0 = empty.clip_records()
        assert r0.shape[0] == 0 and t0.tolist() == [0]


def test_fasta_arrays_equal_the_line_reader(tmp_path):
    """bamio.read_fasta_arrays (no Python loop over the lines) against bamio.read_fasta, odd files included."""
    import os
    from mustache_amd import bamio
    rng = np.random.default_rng(2)
    seq = lambda n: "".join(rng.choice(list("ACGTNacgtn"), n))
    texts = [">c1 some description\n" + "\n".join(seq(60) for _ in range(50)) + "\n>c2\n" + seq(17) + "\n>empty\n>c4\n" + seq(200),   # no newline at the end
             ">w\r\n" + seq(60) + "\r\n" + seq(12) + "\r\n",                                    # CRLF
             ">b\n  " + seq(10) + " \n\t" + seq(5) + "\n\n>c\n" + seq(8) + "\n",                  # blanks: the strip rule
             "", ">only_header", "no header line\n>x\nAC>GT\nA\n"]
    for k, t in enumerate(texts):
        path = os.path.join(tmp_path, "f%d.fa" % k)
        open(path, "w", newline="").write(t)
        want = bamio.read_fasta(path)
        got = bamio.read_fasta_arrays(path)
        assert list(got) == list(want), k
        for name in want:
            assert got[name].tobytes().decode() == want[name], (k, name)


def test_flank_table_from_the_tsv_frame():
    """pairflanks' table of the findflanks TSV: integer columns are taken as they are, anything else value by value (so a
    hole in a count column is an error, as int(nan) is in the reference's row loop), unknown contigs are an error."""
    import io
    import pandas as pd
    from mustache_amd import pairflanks as pfh
    case = util.load_json("sim_a")
    flanks = pd.read_csv(io.StringIO(case["findflanks_tsv"]), sep="\t")
    names = sorted(set(str(c) for c in flanks["contig"]))
    tab, seqs = pfh.flanks_to_table(flanks, names)
    assert len(tab) == len(flanks) > 0
    assert tab["pos"].tolist() == [int(p) for p in flanks["pos"]] and tab["total_count"].tolist() == [int(v) for v in flanks["total_count"]]
    assert tab["orient"].tolist() == [5 if o == "5p" else 3 for o in flanks["orient"]]
    assert bytes(seqs) == "".join(flanks["consensus_seq"]).encode() and int(tab["seq_len"].sum()) == len(seqs)
    as_float = flanks.copy()
    as_float["runthrough_count"] = as_float["runthrough_count"].astype(float)          # whole numbers in a float column: fine
    assert np.array_equal(pfh.flanks_to_table(as_float, names)[0], tab)
    as_float.loc[0, "runthrough_count"] = np.nan
    with pytest.raises(ValueError):
        pfh.flanks_to_table(as_float, names)
    with pytest.raises(KeyError):
        pfh.flanks_to_table(flanks, ["some_other_contig"])

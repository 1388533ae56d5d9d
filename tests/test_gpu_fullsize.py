"""Full-size parity on synthetic BASELINE.json workloads: the CPU oracle finishes these in seconds, so the comparison is
direct (every flank row, pair row, count and sequence), plus size-independent properties."""
import numpy as np
import pytest

from mustache_amd import lib, pairflanks as pfh
from mustache_amd.findflanks import device_reads
from mustache_amd.synth import CONFIGS, synth_packed

pytestmark = pytest.mark.gpu


def _both(hip, oracle, packed, genome):
    import torch
    params = lib.make_params()
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    ranks = pfh.contig_ranks(packed.contig_names)
    rd = device_reads(hip, packed, "cuda:0")
    ff = hip.run_findflanks(rd, params)
    pf = hip.run_pairflanks(rd, params, ff["flanks"], ff["seqs"], ranks, torch.from_numpy(g).to("cuda:0"), goff)
    ro = oracle.reads(packed)
    off = oracle.run_findflanks(ro, params)
    opf = oracle.run_pairflanks(ro, params, off["flanks"], off["seqs"], ranks, g, goff)
    return ff, pf, off, opf


def _same(a, b, dtype, skip=()):
    assert a.shape == b.shape
    for n in dtype.names:
        if n not in skip:
            assert np.array_equal(a[n], b[n]), n


DEEP = dict(contig_lens=(20_000,), coverage=6000, n_insertions=1)   # one site with more clips than a position bucket holds


@pytest.mark.parametrize("name,seed", [("tiny", 3), ("C3_klebsiella", 4), ("C1_ecoli_100x", 5), ("C2_saureus_1000x", 6),
                                       ("C5_tenth", 7), ("deep", 8)])
def test_fullsize_matches_oracle(hip, oracle, name, seed):
    packed, genome, truth = synth_packed(**(DEEP if name == "deep" else CONFIGS[name]), seed=seed)
    ff, pf, off, opf = _both(hip, oracle, packed, genome)
    _same(ff["flanks"], off["flanks"], lib.FLANK_DTYPE)
    assert ff["seqs"] == off["seqs"]
    assert list(ff["stats"][:12]) == list(off["stats"][:12])
    _same(pf["pairs"], opf["pairs"], lib.PAIR_DTYPE)
    assert pf["strs"] == opf["strs"]
    # properties that do not need the oracle: every simulated insertion is recovered as one pair with its TSD
    got = {(packed.contig_names[int(p["contig_id"])], int(p["pos_5p"]), int(p["pos_3p"])) for p in pf["pairs"]}
    assert set(truth) <= got
    pos = ff["flanks"]["pos"].astype(np.int64) + ff["flanks"]["contig_id"].astype(np.int64) * (1 << 32)
    assert np.all(np.diff(pos) >= 0)                                   # rows sorted by (contig, pos)
    tot = sum(ff["flanks"][c].astype(np.int64) for c in (
        "softclip_count_5p", "softclip_count_3p", "runthrough_count", "small_insertion_count_5p",
        "small_insertion_count_3p", "large_insertion_count_5p", "large_insertion_count_3p", "deletion_count"))
    assert np.array_equal(tot, ff["flanks"]["total_count"])            # get_total_count, findflanks.py:789-795


def test_idempotent_and_order_independent_of_pool_extras(hip):
    """Running twice on the same ctx gives identical bytes (workspace reuse)."""
    packed, genome, _ = synth_packed(**CONFIGS["tiny"], seed=9)
    rd = device_reads(hip, packed, "cuda:0")
    a = hip.run_findflanks(rd, lib.make_params())
    b = hip.run_findflanks(rd, lib.make_params())
    assert a["seqs"] == b["seqs"] and a["flanks"].tobytes() == b["flanks"].tobytes()


def test_list_and_device_sort_path_gives_the_same_bytes(hip, monkeypatch):
    """The event keys reach their sorted order through position buckets (LDS sorts) or, when a bucket would overflow,
    through the list + rocPRIM sort: both paths must agree byte for byte."""
    packed, genome, _ = synth_packed(**CONFIGS["tiny"], seed=11)
    rd = device_reads(hip, packed, "cuda:0")
    a = hip.run_findflanks(rd, lib.make_params())
    monkeypatch.setenv("MFK_NO_BUCKET_SORT", "1")
    b = hip.run_findflanks(rd, lib.make_params())
    assert a["seqs"] == b["seqs"] and a["flanks"].tobytes() == b["flanks"].tobytes() and a["sites"].tobytes() == b["sites"].tobytes()
    assert list(a["stats"]) == list(b["stats"]) and len(a["flanks"]) > 0


@pytest.mark.parametrize("name", ["C4_batch12", "C5_meta_50mb"])
def test_fullsize_digest(hip, name):
    """BASELINE configs[3] (the 12 isolates one GPU holds of the 96, as ONE batch) and configs[4] (50 Mb, 200x, 10 000
    insertions: 69 M reads) at full size.  The oracle needs minutes for these, so its results travel as sha256 digests
    (tests/golden/fullsize_digests.json, written by tests/golden/make_fullsize_digests.py in the build container from the
    same seeds); every byte of both result tables and the logged counters go into them."""
    import json
    import torch
    from tests.util import FULLSIZE_DIGESTS, fullsize_inputs, result_digests
    want = json.load(open(FULLSIZE_DIGESTS))
    assert want["numpy"].split(".")[:2] == np.__version__.split(".")[:2], "digests were made with another numpy: the seeded inputs may differ"
    packed, g, goff, ranks = fullsize_inputs(name)
    assert packed.n_reads == want[name]["n_reads"]
    rd = device_reads(hip, packed, "cuda:0")
    params = lib.make_params()
    ff = hip.run_findflanks(rd, params)
    pf = hip.run_pairflanks(rd, params, ff["flanks"], ff["seqs"], ranks, torch.from_numpy(g).to("cuda:0"), goff)
    got = result_digests(ff, pf)
    for k in ("n_flanks", "n_pairs", "stats", "pair_stats", "flanks", "seqs", "pairs", "strs"):
        assert got[k] == want[name][k], k
    # and the fused entry point gives the same bytes at this size
    ff2, pf2 = hip.run_flankpairs(rd, params, ranks, torch.from_numpy(g).to("cuda:0"), goff)
    assert result_digests(ff2, pf2) == got


@pytest.mark.parametrize("kw", [dict(CONFIGS["tiny"]), dict(contig_lens=(400_000,), coverage=200, n_insertions=40),
                                dict(contig_lens=(150_000,), coverage=230, n_insertions=12, clip_error=0.05),
                                dict(contig_lens=(60_000,), coverage=420, n_insertions=6, clip_error=0.02)])
def test_small_job_kernel_and_device_wide_kernels_agree(hip, oracle, monkeypatch, kw):
    """The consensus of a site-side runs in one workgroup (k_job_small: up to 256 words that fit the LDS budget, a second
    launch with a larger budget for those that do not) or through the device-wide kernels; MFK_NO_SMALL_JOBS sends every
    job down the second way.  Same bytes either way, and the oracle's: shallow sites, 200x sites, sites where nearly
    every word is a leaf (5 % errors: the tables outgrow the first budget), sites around the 256-word limit."""
    packed, genome, _ = synth_packed(**kw, seed=13)
    rd = device_reads(hip, packed, "cuda:0")
    params = lib.make_params()
    a = hip.run_findflanks(rd, params)
    monkeypatch.setenv("MFK_NO_SMALL_JOBS", "1")
    b = hip.run_findflanks(rd, params)
    monkeypatch.delenv("MFK_NO_SMALL_JOBS")
    assert a["seqs"] == b["seqs"] and a["flanks"].tobytes() == b["flanks"].tobytes() and list(a["stats"][:12]) == list(b["stats"][:12])
    # k_job_small on characters that k_word_fill wrote to memory first (round 2's flow) instead of making them itself from the pool
    monkeypatch.setenv("MFK_NO_FUSED_FILL", "1")
    c = hip.run_findflanks(rd, params)
    monkeypatch.delenv("MFK_NO_FUSED_FILL")
    assert a["seqs"] == c["seqs"] and a["flanks"].tobytes() == c["flanks"].tobytes() and list(a["stats"][:13]) == list(c["stats"][:13])
    want = oracle.run_findflanks(oracle.reads(packed), params)
    assert a["seqs"] == want["seqs"] and a["flanks"].tobytes() == want["flanks"].tobytes() and len(a["flanks"]) > 0
    assert list(a["stats"][:12]) == list(want["stats"][:12])


@pytest.mark.parametrize("read_len,expect", [(320, "ok"), (531, "ok"), (532, "ok"), (560, "too long")])
def test_long_soft_clips(hip, oracle, read_len, expect):
    """Soft clips of 300, ~510 and exactly 512 bases (250-300 bp and longer libraries; the device-wide consensus kernels then
    ask for up to 64 KB of dynamic LDS on top of their static part) against the oracle; beyond 512 bases the call fails
    with error 6 and a message instead of a launch failure."""
    packed, genome, _ = synth_packed(contig_lens=(60_000,), coverage=60, n_insertions=4, read_len=read_len, frag_mean=read_len + 100, seed=3)
    f = packed.flags
    longest = int(max(np.where((f & 3) == 1, packed.first_op_len, 0).max(), np.where(((f >> 2) & 3) == 1, packed.last_op_len, 0).max()))
    rd = device_reads(hip, packed, "cuda:0")
    params = lib.make_params()
    if expect == "too long":
        assert longest > 512
        with pytest.raises(lib.MfkError, match="error 6.*longer than 512"):
            hip.run_findflanks(rd, params)
        # the call failed half-way, with a fill still queued on a side stream: the next call on the same ctx is unharmed
        small, sg, _ = synth_packed(**CONFIGS["tiny"], seed=21)
        ff, pf, off, opf = _both(hip, oracle, small, sg)
        assert ff["flanks"].tobytes() == off["flanks"].tobytes() and ff["seqs"] == off["seqs"] and pf["strs"] == opf["strs"] and len(ff["flanks"]) > 0
        return
    assert {320: 300, 531: 508, 532: 512}[read_len] <= longest <= 512
    ff, pf, off, opf = _both(hip, oracle, packed, genome)
    _same(ff["flanks"], off["flanks"], lib.FLANK_DTYPE)
    assert ff["seqs"] == off["seqs"] and len(ff["flanks"]) > 0 and int(ff["flanks"]["seq_len"].max()) >= longest - 8
    _same(pf["pairs"], opf["pairs"], lib.PAIR_DTYPE)
    assert pf["strs"] == opf["strs"]
    # and through the device-wide kernels alone (the small-job kernel takes words up to 255 characters only)
    import os
    os.environ["MFK_NO_SMALL_JOBS"] = "1"
    try:
        b = hip.run_findflanks(rd, params)
    finally:
        del os.environ["MFK_NO_SMALL_JOBS"]
    assert b["seqs"] == off["seqs"] and b["flanks"].tobytes() == off["flanks"].tobytes()


def test_documented_limits_fail_with_messages(hip, oracle):
    """more than 65535 contigs, a reference beyond the 32-bit site axis and max_direct_repeat_length beyond
    65000 are rejected with a message before any kernel runs, and the ctx works afterwards."""
    packed, genome, _ = synth_packed(**CONFIGS["tiny"], seed=22)
    tensors = packed.to_torch("cuda:0")
    hdr = lib.host_header(packed)
    params = lib.make_params()
    n = packed.n_reads
    # 70 000 contigs (the reads sit on the first)
    many = dict(hdr, contig_len=np.full(70_000, 1000, dtype=np.int32), contig_read_start=np.concatenate([[0], np.full(70_000, n)]).astype(np.int64))
    with pytest.raises(lib.MfkError, match="n_contigs 70000 out of range"):
        hip.run_findflanks(hip.reads_struct(tensors, n, packed.max_span, many), params)
    # 6.3 Gb of reference (the flat site axis is 32 bits wide)
    big = dict(hdr, contig_len=np.array([int(hdr["contig_len"][0]), 2_100_000_000, 2_100_000_000, 2_100_000_000], dtype=np.int32),
               contig_read_start=np.array([0, n, n, n, n], dtype=np.int64))
    with pytest.raises(lib.MfkError, match="does not fit the 32-bit site axis"):
        hip.run_findflanks(hip.reads_struct(tensors, n, packed.max_span, big), params)
    import torch
    ff = hip.run_findflanks(device_reads(hip, packed, "cuda:0"), params)
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    with pytest.raises(lib.MfkError, match="max_direct_repeat_length out of range"):
        hip.run_pairflanks(device_reads(hip, packed, "cuda:0"), lib.make_params(max_direct_repeat_length=70_000), ff["flanks"], ff["seqs"],
                           pfh.contig_ranks(packed.contig_names), torch.from_numpy(g).to("cuda:0"), goff)
    # unharmed
    a, pf, off, opf = _both(hip, oracle, packed, genome)
    assert a["flanks"].tobytes() == off["flanks"].tobytes() and pf["strs"] == opf["strs"]


@pytest.mark.parametrize("pf_kw,what", [(dict(), "thousands of candidate pairs: pair rows made on the device"),
                                        (dict(max_direct_repeat_length=3), "a few hundred candidates: pair rows made by the host loop"),
                                        (dict(max_direct_repeat_length=0), "no candidate pair at all"),
                                        (dict(min_softclip_ratio=0.995), "nearly every site dies in the ratio filter: a handful of flank rows")])
def test_fused_call_hands_the_flank_table_over_on_the_device(hip, monkeypatch, pf_kw, what):
    """mfk_flankpairs with thousands of active sites: findflanks leaves its rows on the device, the pairing stage reads them there
    and brings every table back with its own read-backs.  Same bytes and counters as the two separate calls (which go through
    the host table), and as the fused call with the hand-over switched off."""
    import torch
    packed, genome, truth = synth_packed(contig_lens=(6_000_000,), coverage=40, n_insertions=2600, seed=31)
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    ranks = pfh.contig_ranks(packed.contig_names)
    rd = device_reads(hip, packed, "cuda:0")
    gd = torch.from_numpy(g).to("cuda:0")
    params = lib.make_params(**pf_kw)
    ff = hip.run_findflanks(rd, params)
    pf = hip.run_pairflanks(rd, params, ff["flanks"], ff["seqs"], ranks, gd, goff)
    assert int(ff["stats"][3]) > 2048, "the case is meant to have more than 2048 active sites"
    ff2, pf2 = hip.run_flankpairs(rd, params, ranks, gd, goff)
    monkeypatch.setenv("MFK_NO_HANDOFF", "1")
    ff3, pf3 = hip.run_flankpairs(rd, params, ranks, gd, goff)
    monkeypatch.delenv("MFK_NO_HANDOFF")
    for a, b in ((ff2, pf2), (ff3, pf3)):
        assert a["sites"].tobytes() == ff["sites"].tobytes() and a["flanks"].tobytes() == ff["flanks"].tobytes() and a["seqs"] == ff["seqs"]
        assert list(a["stats"][:13]) == list(ff["stats"][:13])
        assert b["pairs"].tobytes() == pf["pairs"].tobytes() and b["strs"] == pf["strs"] and list(b["stats"][:5]) == list(pf["stats"][:5])
    if not pf_kw:
        assert len(pf["pairs"]) > 2000
    if pf_kw.get("max_direct_repeat_length") == 0:
        assert len(pf["pairs"]) == 0 and len(ff["flanks"]) > 4000
    # assign_pairs per component of candidates that share flanks (k_assign_seg) against the device-wide sort + one-workgroup rounds
    monkeypatch.setenv("MFK_NO_SEG_ASSIGN", "1")
    pf5 = hip.run_pairflanks(rd, params, ff["flanks"], ff["seqs"], ranks, gd, goff)
    monkeypatch.delenv("MFK_NO_SEG_ASSIGN")
    assert pf5["pairs"].tobytes() == pf["pairs"].tobytes() and pf5["strs"] == pf["strs"] and list(pf5["stats"][:5]) == list(pf["stats"][:5])
    # a second fused call right behind (the arena is reused): same again
    ff4, pf4 = hip.run_flankpairs(rd, params, ranks, gd, goff)
    assert ff4["flanks"].tobytes() == ff["flanks"].tobytes() and pf4["pairs"].tobytes() == pf["pairs"].tobytes() and pf4["strs"] == pf["strs"]


@pytest.mark.parametrize("name,seed,minial", [("tiny", 31, 21), ("C3_klebsiella", 32, 21), ("C2_saureus_1000x", 33, 21),
                                              ("deep", 34, 21), ("C3_klebsiella", 35, 300), ("tiny", 36, 0)])
def test_dense_clip_records_change_nothing(hip, name, seed, minial):
    """include/mfk.h mfk_clip_rec: the optional dense copy of the clipped reads is a layout, not an input -- with it, without
    it, and with -minial beyond what a record's `inner` byte holds (the kernel then takes the lengths from the columns)
    every stage gives the same bytes."""
    import torch
    packed, genome, _ = synth_packed(**(DEEP if name == "deep" else CONFIGS[name]), seed=seed)
    params = lib.make_params(min_alignment_inner_length=minial)
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    ranks = pfh.contig_ranks(packed.contig_names)
    gd = torch.from_numpy(g).to("cuda:0")
    with_rec, without = device_reads(hip, packed, "cuda:0"), device_reads(hip, packed, "cuda:0", clip_rec=False)
    assert with_rec.clip_rec and with_rec.n_clip == int(np.count_nonzero(packed.a["flags"] & 15)) and not without.clip_rec
    a, ap = hip.run_flankpairs(with_rec, params, ranks, gd, goff)
    b, bp = hip.run_flankpairs(without, params, ranks, gd, goff)
    _same(a["flanks"], b["flanks"], lib.FLANK_DTYPE)
    assert a["seqs"] == b["seqs"] and list(a["stats"][:12]) == list(b["stats"][:12])
    _same(ap["pairs"], bp["pairs"], lib.PAIR_DTYPE)
    assert ap["strs"] == bp["strs"]
    for stage in (1, 2):                                   # the unfiltered / filtered site tables too
        sa, sb = hip.run_findflanks(with_rec, params, stop_after=stage), hip.run_findflanks(without, params, stop_after=stage)
        _same(sa["sites"], sb["sites"], lib.SITE_DTYPE)


def test_clip_records_without_their_tile_table_are_refused(hip):
    packed, _, _ = synth_packed(**CONFIGS["tiny"], seed=37)
    rd = device_reads(hip, packed, "cuda:0")
    rd.clip_tile = None
    with pytest.raises(lib.MfkError, match="clip_tile"):
        hip.run_findflanks(rd, lib.make_params())

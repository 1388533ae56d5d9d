"""GPU parity tests: the HIP library (through its C-ABI) against the CPU oracle and the golden vectors of the reference.
Bit-exact: every count, position, flag and sequence must be identical."""
import numpy as np
import pytest

from mustache_amd import lib
from mustache_amd import pairflanks as pfh
from mustache_amd.findflanks import device_reads
from mustache_amd.pack import pack_reads
from mustache_amd.simulate import simulate_readset
from tests import util

pytestmark = pytest.mark.gpu


def _device_reads(hip, packed):
    from mustache_amd.findflanks import device_reads
    return device_reads(hip, packed, "cuda:0")


@pytest.mark.parametrize("name", util.ALL_CASES)
@pytest.mark.parametrize("small_jobs", [True, False])
def test_golden_tsvs_bit_exact(hip, name, small_jobs, monkeypatch):
    """findflanks + pairflanks TSVs equal to what the reference printed (IR columns: build's definition), with the consensus
    through k_job_small where a job fits it and through the device-wide kernels for every job (MFK_NO_SMALL_JOBS).  The
    deep_* cases (tests/golden/make_golden_deep.py) hold hundreds of leaves and several clusters per site-side, IUPAC clip
    bases, clips past 200 bases and two surviving consensus sequences per side."""
    import torch
    if not small_jobs:
        monkeypatch.setenv("MFK_NO_SMALL_JOBS", "1")
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    rsd = _device_reads(hip, packed)
    ff, stats, pf = util.run_both_stages(hip, rsd, packed, case, genome_arr=lambda g: torch.from_numpy(g).to("cuda:0"))
    assert ff == case["findflanks_tsv"]
    assert stats[1:10] == case["findflanks_log_counts"]
    assert pf == case["pairflanks_tsv"]


@pytest.mark.parametrize("name", ["quirks", "sim_a", "sim_c"])
@pytest.mark.parametrize("stage", range(1, 10))
def test_site_table_after_every_stage(hip, oracle, name, stage):
    """Site table (positions, ten counts, keep flags) after each call of _findflanks' sequence, vs the oracle."""
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="clipped")
    params = lib.make_params(**case["findflanks_params"])
    got = hip.run_findflanks(_device_reads(hip, packed), params, stop_after=stage)
    want = oracle.run_findflanks(oracle.reads(packed), params, stop_after=stage)
    util.assert_same_sites(got["sites"], want["sites"])
    assert list(got["stats"][:12]) == list(want["stats"][:12])


@pytest.mark.parametrize("seed", range(20, 32))
def test_random_simulations_match_oracle(hip, oracle, seed):
    import torch
    rng = np.random.default_rng(seed)
    kw = dict(seed=seed, contig_lens=(int(rng.integers(3000, 9000)),) * int(rng.integers(1, 4)),
              n_insertions=int(rng.integers(1, 4)), coverage=int(rng.integers(10, 120)),
              read_len=int(rng.choice([75, 100, 150])), sub_rate=float(rng.choice([0.002, 0.01, 0.03])),
              wildtype_frac=float(rng.choice([0, 0.1, 0.4])))
    fp = dict(min_softclip_length=int(rng.choice([4, 8])), min_softclip_count=int(rng.choice([2, 4])),
              min_softclip_ratio=float(rng.choice([0.1, 0.15])), min_count_consensus=int(rng.choice([1, 2, 3])),
              max_indel_ratio=float(rng.choice([0.03, 0.2])))
    genome, rs, _ = simulate_readset(**kw)
    packed = pack_reads(rs.contigs, rs, pool="all")
    case = {"findflanks_params": fp, "pairflanks_params": {}, "genome": genome}
    o_res = oracle.run_findflanks(oracle.reads(packed), lib.make_params(**fp))
    # pairflanks consumes the findflanks TSV: feed both backends the oracle's table
    from mustache_amd.findflanks import flank_frame
    df = flank_frame(o_res, packed.contig_names)
    df.insert(0, "flank_id", list(df.index + 1))
    df.insert(0, "sample", "sample1")
    case["findflanks_tsv"] = df.to_csv(sep="\t", index=False)
    want = util.run_both_stages(oracle, oracle.reads(packed), packed, case)
    got = util.run_both_stages(hip, _device_reads(hip, packed), packed, case,
                               genome_arr=lambda g: torch.from_numpy(g).to("cuda:0"))
    assert got[0] == want[0]
    assert got[1][:12] == want[1][:12]
    assert got[2] == want[2]


def test_ir_detect_matches_oracle(hip, oracle):
    rng = np.random.default_rng(7)
    seqs = []
    for k in range(300):
        n5, n3 = int(rng.integers(8, 41)), int(rng.integers(8, 41))
        a = "".join("ACGT"[i] for i in rng.integers(0, 4, n5))
        if k % 3:
            comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
            arm = "".join(comp[c] for c in reversed(a[:int(rng.integers(6, n5 + 1))]))
            b = "".join("ACGT"[i] for i in rng.integers(0, 4, max(0, n3 - len(arm)))) + arm
            b = list(b)
            for _ in range(int(rng.integers(0, 3))):
                b[int(rng.integers(0, len(b)))] = "ACGTN"[int(rng.integers(0, 5))]
            b = "".join(b)[-40:]
        else:
            b = "".join("ACGT"[i] for i in rng.integers(0, 4, n3))
        seqs.append(a + "N" * 20 + b)
    got = hip.run_ir_detect(seqs)
    want = oracle.run_ir_detect(seqs)
    assert np.array_equal(got, want)
    assert want[:, 0].sum() > 50


def _ir_seqs(seed, n=300):
    rng = np.random.default_rng(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    seqs = []
    for k in range(n):
        n5, n3 = int(rng.integers(8, 41)), int(rng.integers(8, 41))
        a = "".join("ACGT"[i] for i in rng.integers(0, 4, n5))
        if k % 3:
            arm = "".join(comp[c] for c in reversed(a[:int(rng.integers(6, n5 + 1))]))
            b = list("".join("ACGT"[i] for i in rng.integers(0, 4, max(0, n3 - len(arm)))) + arm)
            for _ in range(int(rng.integers(0, 3))):
                b[int(rng.integers(0, len(b)))] = "ACGTN"[int(rng.integers(0, 5))]
            b = "".join(b)[-40:]
        else:
            b = "".join("ACGT"[i] for i in rng.integers(0, 4, n3))
        seqs.append(a + "N" * 20 + b)
    return seqs


def test_reported_inverted_repeats_are_self_consistent(hip):
    """The IR columns have no reference output to be pinned against (EMBOSS einverted is in neither tree nor image), so
    beside the comparison with the oracle's model this checks what any einverted-like detector must satisfy, with no
    oracle involved: the two arms are disjoint and ordered, they lie where mustache accepts them (arm 1 starts within
    15 bases of the start, arm 2 ends within 15 of the end, pairflanks.py:253-266), and arm 1 aligned against the
    reverse complement of arm 2 scores at least the threshold 15 under match +3 / mismatch -4 / gap -12
    (embosstools.py:8-14)."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    seqs = _ir_seqs(19)
    got = hip.run_ir_detect(seqs)
    n_ir = 0
    for s, (has, a1s, a1e, a2s, a2e, nrep) in zip(seqs, got):
        if not has:
            continue
        n_ir += 1
        L = len(s)
        assert 1 <= a1s <= a1e < a2s <= a2e <= L
        assert a1s <= 15 and a2e >= L - 15
        x, y = s[a1s - 1:a1e], s[a2s - 1:a2e][::-1]
        best = np.full((len(x) + 1, len(y) + 1), -10 ** 6, dtype=np.int64)    # global alignment of the two arms
        best[0, :] = -12 * np.arange(len(y) + 1)
        best[:, 0] = -12 * np.arange(len(x) + 1)
        for i in range(1, len(x) + 1):
            for j in range(1, len(y) + 1):
                m = 3 if comp.get(x[i - 1]) == y[j - 1] else -4
                best[i, j] = max(best[i - 1, j - 1] + m, best[i - 1, j] - 12, best[i, j - 1] - 12)
        assert best[len(x), len(y)] >= 15, (s, a1s, a1e, a2s, a2e)
        assert comp.get(x[0]) == y[0] and comp.get(x[-1]) == y[-1]          # a local alignment starts and ends on a pair
    assert n_ir > 50


def test_empty_and_tiny_inputs(hip, oracle):
    from mustache_amd.records import AlignedRead, ReadSet
    reads = [AlignedRead("a", "c1", 10, 60, [(0, 50)], "A" * 50, [30] * 50, md="50")]
    rs = ReadSet([("c1", 1000)], reads)
    packed = pack_reads(rs.contigs, rs, pool="all")
    res = hip.run_findflanks(_device_reads(hip, packed), lib.make_params())
    assert len(res["flanks"]) == 0 and len(res["sites"]) == 0
    # no read at all, with and without the (empty) dense clip records
    none = pack_reads(rs.contigs, ReadSet([("c1", 1000)], []), pool="all")
    assert none.n_reads == 0 and none.clip_records()[1].tolist() == [0]
    for clip_rec in (True, False):
        res = hip.run_findflanks(device_reads(hip, none, "cuda:0", clip_rec=clip_rec), lib.make_params())
        assert len(res["flanks"]) == 0 and len(res["sites"]) == 0


@pytest.mark.parametrize("name", util.CASES)
def test_recall_golden_bit_exact(hip, name):
    """mfk_recall through the C-ABI equals what the reference's recall.py wrote."""
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="none")
    rsd = _device_reads(hip, packed)
    for e in util.load_json("recall_vectors")[name]:
        assert util.recall_text(hip, rsd, packed, e) == e["recall_tsv"], e["label"]


@pytest.mark.parametrize("seed", (41, 42, 43))
def test_recall_random_sites_match_oracle(hip, oracle, seed):
    rng = np.random.default_rng(seed)
    genome, rs, _ = simulate_readset(seed=seed, contig_lens=(6000, 3000), n_insertions=3, coverage=int(rng.integers(20, 200)),
                                     read_len=100, sub_rate=0.01, wildtype_frac=0.2)
    packed = pack_reads(rs.contigs, rs, pool="none")
    cid = rng.integers(0, 2, 400).astype(np.int32)
    pos = np.array([int(rng.integers(-1, packed.contig_len[c] + 1)) for c in cid], dtype=np.int32)
    params = lib.make_params(min_alignment_quality=int(rng.choice([0, 20, 40])), large_insertion_cutoff=int(rng.choice([3, 30])))
    got = hip.run_recall(_device_reads(hip, packed), params, cid, pos)
    want = oracle.run_recall(oracle.reads(packed), params, cid, pos)
    assert np.array_equal(got, want)
    assert want.sum() > 0
    with pytest.raises(lib.MfkError):
        hip.run_recall(_device_reads(hip, packed), params, [0], [int(packed.contig_len[0]) + 1])


def test_overlap_golden_bit_exact(hip, oracle):
    """mfk_overlap through the C-ABI: the reference's tuples on 160 string pairs and its TSVs on nine pairs tables."""
    vec = util.load_json("overlap_vectors")
    got = hip.run_overlap([v[0] for v in vec["sliding"]], [v[1] for v in vec["sliding"]])
    assert [[int(x) for x in row] for row in got] == [v[2] for v in vec["sliding"]]
    for e in vec["tables"]:
        assert util.overlap_text(hip, e) == e["overlap_tsv"], e["label"]
    # long sequences (past the LDS staging limit) and many pairs, against the oracle
    rng = np.random.default_rng(7)
    def rs(n, a="ACGT"):
        return "".join(a[i] for i in rng.integers(0, len(a), n))
    s5 = [rs(int(rng.integers(1, 400))) for _ in range(500)] + [rs(30000), rs(20000, "AC"), rs(5)]
    s3 = [rs(int(rng.integers(1, 400))) for _ in range(500)] + [rs(10000), rs(21000, "AC"), rs(40000)]
    s3[0] = s5[0][-20:] + s3[0]
    assert np.array_equal(hip.run_overlap(s5, s3), oracle.run_overlap(s5, s3))
    with pytest.raises(lib.MfkError):
        hip.run_overlap(["ACGT"], [""])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sim_a", "sim_c", "quirks"])
def test_flankpairs_one_call_equals_two(hip, name):
    """mfk_flankpairs = mfk_findflanks followed by mfk_pairflanks on its table: the same bytes from both."""
    import torch
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    R = device_reads(hip, packed, "cuda:0")
    params = lib.make_params(**dict(case["findflanks_params"], **case["pairflanks_params"]))
    g, goff = pfh.genome_arrays(case["genome"], packed.contig_names)
    gd = torch.from_numpy(g).to("cuda:0")
    ranks = pfh.contig_ranks(packed.contig_names)
    ff = hip.run_findflanks(R, params)
    pf = hip.run_pairflanks(R, params, ff["flanks"], ff["seqs"], ranks, gd, goff)
    ff2, pf2 = hip.run_flankpairs(R, params, ranks, gd, goff)
    assert ff2["flanks"].tobytes() == ff["flanks"].tobytes() and ff2["seqs"] == ff["seqs"] and ff2["sites"].tobytes() == ff["sites"].tobytes()
    assert list(ff2["stats"]) == list(ff["stats"])
    assert pf2["pairs"].tobytes() == pf["pairs"].tobytes() and pf2["strs"] == pf["strs"] and list(pf2["stats"]) == list(pf["stats"])
    assert len(pf["pairs"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sim_b", "quirks"])
def test_pair_window_device_path(hip, name, monkeypatch):
    """Small flank tables list their candidate pairs on the host; MFK_NO_HOST_PAIRS forces the kernels that large tables
    use (k_pair_window + scan + rocPRIM sorts need > 256 flanks otherwise).  Same pairs table either way."""
    import torch
    case = util.load_json(name)
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    R = device_reads(hip, packed, "cuda:0")
    params = lib.make_params(**dict(case["findflanks_params"], **case["pairflanks_params"]))
    g, goff = pfh.genome_arrays(case["genome"], packed.contig_names)
    gd = torch.from_numpy(g).to("cuda:0")
    ranks = pfh.contig_ranks(packed.contig_names)
    ff, pf = hip.run_flankpairs(R, params, ranks, gd, goff)
    monkeypatch.setenv("MFK_NO_HOST_PAIRS", "1")
    ff2, pf2 = hip.run_flankpairs(R, params, ranks, gd, goff)
    assert pf2["pairs"].tobytes() == pf["pairs"].tobytes() and pf2["strs"] == pf["strs"] and list(pf2["stats"]) == list(pf["stats"])
    assert len(pf["pairs"]) > 0


@pytest.mark.gpu
def test_chained_scan_epoch_wrap(hip, oracle, monkeypatch):
    """More than 16 k events: the prefix sums go through k_scan_chain, whose tile entries are tagged with a 16-bit call
    counter.  A ctx that starts just below the wrap (test hook MFK_SCAN_EPOCH0) crosses it within three calls; every call
    must give the oracle's tables."""
    import torch
    from mustache_amd.synth import synth_packed
    packed, genome, _ = synth_packed(contig_lens=(2_400_000,), coverage=150, n_insertions=6, seed=77)
    params = lib.make_params()
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    ranks = pfh.contig_ranks(packed.contig_names)
    Ro = oracle.reads(packed)
    want = oracle.run_findflanks(Ro, params)
    wantp = oracle.run_pairflanks(Ro, params, want["flanks"], want["seqs"], ranks, g, goff)
    monkeypatch.setenv("MFK_SCAN_EPOCH0", str(0xFFFF - 5))
    own = lib.Binding(lib.LIB_PATH, "mfk_", 0)
    try:
        R = device_reads(own, packed, "cuda:0")
        gd = torch.from_numpy(g).to("cuda:0")
        for _ in range(4):
            ff, pf = own.run_flankpairs(R, params, ranks, gd, goff)
            assert int(ff["stats"][10]) > 16 * 1024          # events: beyond the one-workgroup scan
            assert ff["flanks"].tobytes() == want["flanks"].tobytes() and ff["seqs"] == want["seqs"]
            assert pf["pairs"].tobytes() == wantp["pairs"].tobytes() and pf["strs"] == wantp["strs"]
        assert len(want["flanks"]) >= 10
    finally:
        own.close()


@pytest.mark.gpu
def test_library_contexts_in_concurrent_threads_give_the_same_tables(hip):
    """Two more contexts on the same GPU, a host thread each, all three running the fused call on the same resident inputs at
    once (a cohort run with several samples per GPU): every context's tables equal the single-context ones."""
    import torch
    import bench
    from mustache_amd import lib as _lib, pairflanks as pfh
    from mustache_amd.findflanks import device_reads
    from mustache_amd.synth import synth_packed
    packed, genome, truth = synth_packed(contig_lens=(120000, 30000), n_insertions=12, coverage=120, seed=77, pool="all")
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    reads = device_reads(hip, packed, "cuda:0")
    g_dev = torch.from_numpy(g).to("cuda:0")
    r = bench.concurrent_contexts(hip, reads, _lib.make_params(), pfh.contig_ranks(packed.contig_names), g_dev, goff, packed.n_reads, 3, 15, 0)
    assert r["tables_equal"] and not r["errors"] and r["steps"] == 45


@pytest.mark.gpu
def test_samples_over_concurrent_contexts_equal_their_own_runs(hip):
    """cohort.flankpairs_concurrent: seven samples of different size over three contexts, results in sample order and equal
    to each sample's own run; a failing sample stops the queue and its error reaches the caller."""
    from mustache_amd import cohort
    from mustache_amd.synth import synth_packed
    samples = []
    for k in range(7):
        packed, genome, truth = synth_packed(contig_lens=(40000 + 9000 * k, 12000), n_insertions=3 + k, coverage=40 + 10 * k, seed=300 + k, pool="all")
        g, goff = pfh.genome_arrays(genome, packed.contig_names)
        samples.append((packed, g, goff))
    params = lib.make_params()
    got = cohort.flankpairs_concurrent(samples, params, "cuda:0", contexts=3)
    assert len(got) == 7
    import torch
    for (packed, g, goff), (ff, pf) in zip(samples, got):
        reads = device_reads(hip, packed, "cuda:0")
        wf, wp = hip.run_flankpairs(reads, params, pfh.contig_ranks(packed.contig_names), torch.from_numpy(g).to("cuda:0"), goff)
        assert len(wf["flanks"]) > 0
        assert np.array_equal(ff["flanks"], wf["flanks"]) and ff["seqs"] == wf["seqs"]
        assert np.array_equal(pf["pairs"], wp["pairs"]) and pf["strs"] == wp["strs"]
    bad = list(samples)
    bad[2] = (None, samples[2][1], samples[2][2])                      # (fails on the host, before anything reaches the device)
    with pytest.raises(AttributeError):
        cohort.flankpairs_concurrent(bad, params, "cuda:0", contexts=3)
    again = cohort.flankpairs_concurrent(samples[:2], params, "cuda:0", contexts=3)      # the contexts are usable afterwards
    assert np.array_equal(again[1][0]["flanks"], got[1][0]["flanks"])

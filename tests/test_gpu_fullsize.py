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

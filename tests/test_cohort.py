"""Multi-isolate batches (BASELINE config C4): the isolates of a rank run as ONE findflanks / pairflanks call.  Each
sample's rows must be exactly those of its own run.  CPU: through the oracle (it takes the same packed arrays) and the
gloo all-gather of the cohort table; GPU: through the HIP library, against the per-sample oracle runs."""
import io
import os
import socket

import numpy as np
import pandas as pd
import pytest

from mustache_amd import cohort, lib
from mustache_amd import pairflanks as pfh
from mustache_amd.findflanks import device_reads, flank_frame
from mustache_amd.pack import pack_reads
from tests import util

BATCH_CASES = ["sim_a", "quirks", "sim_c", "sim_b", "sim_a"]     # a case twice: equal contig names in two samples


def _packs():
    cases = [util.load_json(n) for n in BATCH_CASES]
    packs = []
    for c in cases:
        rs = util.case_readset(c)
        packs.append(pack_reads(rs.contigs, rs, pool="all"))
    return cases, packs


def _same_rows(a, b, sa, sb, names):
    assert len(a) == len(b)
    for n in names:
        if not n.endswith("_off"):
            assert np.array_equal(a[n], b[n]), n
    return True


def _check_batch(binding, reads_of, genome_of, oracle):
    """Batch through `binding` vs every sample alone through the oracle, findflanks then pairflanks (golden params)."""
    cases, packs = _packs()
    batch = cohort.concat_samples(packs, ["s%d" % i for i in range(len(packs))])
    fparams = lib.make_params(**cases[0]["findflanks_params"])
    pparams = lib.make_params(**cases[0]["pairflanks_params"])
    R = reads_of(batch.packed)
    per_sample = cohort.findflanks_batch(binding, batch, R, fparams)
    assert len(per_sample) == len(packs)
    g, goff = cohort.batch_genome(batch, [c["genome"] for c in cases])
    pairs = cohort.pairflanks_batch(binding, batch, R, pparams, per_sample, genome_of(g), goff)
    n_rows = n_pairs = 0
    for s, (case, p) in enumerate(zip(cases, packs)):
        Ro = oracle.reads(p)
        want = oracle.run_findflanks(Ro, fparams)
        got = per_sample[s]
        _same_rows(got["flanks"], want["flanks"], got["seqs"], want["seqs"], lib.FLANK_DTYPE.names)
        assert flank_frame(got, p.contig_names).equals(flank_frame(want, p.contig_names))
        gs, gso = pfh.genome_arrays(case["genome"], p.contig_names)
        wantp = oracle.run_pairflanks(Ro, pparams, want["flanks"], want["seqs"], pfh.contig_ranks(p.contig_names), gs, gso)
        a = pfh.pair_frame(pairs[s], got["flanks"], got["seqs"], p.contig_names)
        b = pfh.pair_frame(wantp, want["flanks"], want["seqs"], p.contig_names)
        assert a.to_csv(sep="\t", index=False) == b.to_csv(sep="\t", index=False)
        n_rows += len(want["flanks"]); n_pairs += len(wantp["pairs"])
    assert n_rows > 10 and n_pairs > 5


def test_batch_equals_per_sample_runs_oracle(oracle):
    _check_batch(oracle, oracle.reads, lambda g: g, oracle)


def test_concat_layout():
    _, packs = _packs()
    batch = cohort.concat_samples(packs)
    m = batch.packed
    assert m.n_reads == sum(p.n_reads for p in packs)
    assert list(batch.contig_start) == list(np.cumsum([0] + [len(p.contig_names) for p in packs]))
    assert np.all(np.diff(m.contig_read_start) >= 0) and m.contig_read_start[-1] == m.n_reads
    # links stay inside their sample and point backwards
    r0 = 0
    for p in packs:
        l = m.link[r0:r0 + p.n_reads]
        assert np.all((l == -1) | ((l >= r0) & (l < r0 + np.arange(p.n_reads))))
        r0 += p.n_reads
    # every pooled read's sequence survives the move
    r0 = 0
    for p in packs:
        for r in (0, p.n_reads // 2, p.n_reads - 1):
            assert m.pool_sequence(r0 + r) == p.pool_sequence(r)
        r0 += p.n_reads
    assert np.array_equal(batch.sample_of_contig(np.arange(batch.contig_start[-1])),
                          np.repeat(np.arange(len(packs)), [len(p.contig_names) for p in packs]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from mustache_amd.dist import shard_samples
    ora = Oracle()
    cases, packs = _packs()
    mine = shard_samples(len(packs), world, rank)
    batch = cohort.concat_samples([packs[i] for i in mine], ["s%d" % i for i in mine])
    params = lib.make_params(**cases[0]["findflanks_params"])
    per_sample = cohort.findflanks_batch(ora, batch, ora.reads(batch.packed), params)
    # rows travel with the global sample number: rank r owns samples r, r + world, ...
    out_rows, out_blob = cohort.cohort_table(per_sample, sample_numbers=mine, device="cpu")
    if rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), out_rows)
        with open(os.path.join(out_dir, "seqs.bin"), "wb") as fh:
            fh.write(out_blob)
    dist.destroy_process_group()


def test_cohort_table_allgather_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp
    from mustache_amd.dist import shard_samples
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rows = np.load(os.path.join(tmp_path, "rows.npy"))
    blob = open(os.path.join(tmp_path, "seqs.bin"), "rb").read()
    cases, packs = _packs()
    params = lib.make_params(**cases[0]["findflanks_params"])
    want_rows, want_seqs = [], []
    for rank in range(world):                       # rank order, then the rank's samples in its batch order
        for k, i in enumerate(shard_samples(len(packs), world, rank)):
            w = oracle.run_findflanks(oracle.reads(packs[i]), params)
            for r in w["flanks"]:
                want_rows.append((i, int(r["contig_id"]), int(r["pos"]), int(r["orient"]), int(r["consensus_softclip_count"])))
                want_seqs.append(w["seqs"][int(r["seq_off"]):int(r["seq_off"]) + int(r["seq_len"])])
    got_rows = [(int(r["contig_id"]) >> 16, int(r["contig_id"]) & 0xFFFF, int(r["pos"]), int(r["orient"]),
                 int(r["consensus_softclip_count"])) for r in rows]
    got_seqs = [blob[int(r["seq_off"]):int(r["seq_off"]) + int(r["seq_len"])] for r in rows]
    assert got_rows == want_rows and got_seqs == want_seqs and len(got_rows) > 10


@pytest.mark.gpu
def test_batch_equals_per_sample_runs_hip(hip, oracle):
    import torch
    _check_batch(hip, lambda packed: device_reads(hip, packed, "cuda:0"), lambda g: torch.from_numpy(g).to("cuda:0"), oracle)


@pytest.mark.gpu
def test_synthetic_isolates_batch_hip(hip, oracle):
    """Six simulated isolates (different seeds) as one batch on the GPU vs six oracle runs."""
    import torch
    from mustache_amd.synth import synth_packed
    packs, genomes = [], []
    for seed in range(6):
        p, genome, _ = synth_packed(contig_lens=(120_000, 30_000), coverage=80, n_insertions=5, seed=40 + seed)
        packs.append(p); genomes.append(genome)
    batch = cohort.concat_samples(packs)
    params = lib.make_params()
    R = device_reads(hip, batch.packed, "cuda:0")
    per_sample = cohort.findflanks_batch(hip, batch, R, params)
    g, goff = cohort.batch_genome(batch, genomes)
    pairs = cohort.pairflanks_batch(hip, batch, R, params, per_sample, torch.from_numpy(g).to("cuda:0"), goff)
    total = 0
    for s, p in enumerate(packs):
        Ro = oracle.reads(p)
        want = oracle.run_findflanks(Ro, params)
        _same_rows(per_sample[s]["flanks"], want["flanks"], None, None, lib.FLANK_DTYPE.names)
        assert flank_frame(per_sample[s], p.contig_names).equals(flank_frame(want, p.contig_names))
        gs, gso = pfh.genome_arrays(genomes[s], p.contig_names)
        wantp = oracle.run_pairflanks(Ro, params, want["flanks"], want["seqs"], pfh.contig_ranks(p.contig_names), gs, gso)
        a = pfh.pair_frame(pairs[s], per_sample[s]["flanks"], per_sample[s]["seqs"], p.contig_names)
        b = pfh.pair_frame(wantp, want["flanks"], want["seqs"], p.contig_names)
        assert a.to_csv(sep="\t", index=False) == b.to_csv(sep="\t", index=False)
        total += len(wantp["pairs"])
    assert total >= 12

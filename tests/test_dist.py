"""Multi-process layer on CPU: gloo, world_size 2.  Sharding must not change results: every rank runs the two sharded
stages (`findflanks_sharded`, `pairflanks_sharded`: the functions bench.py --sharding and the GPU suite drive, here with
the CPU oracle behind the same Binding signatures), the all-gathers merge them, and both tables equal the unsharded run."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mustache_amd import dist as mdist
from mustache_amd import lib, pairflanks as pfh
from mustache_amd.pack import pack_reads
from mustache_amd.synth import synth_packed
from tests import util

MULTI = dict(contig_lens=(300_000, 30_000, 8_000, 120_000), coverage=60, n_insertions=40)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs(case_name):
    if case_name == "multi":
        packed, genome, _ = synth_packed(**MULTI, seed=3)
        params = lib.make_params()
    else:
        case = util.load_json(case_name)
        rs = util.case_readset(case)
        packed = pack_reads(rs.contigs, rs, pool="all")
        genome = case["genome"]
        params = lib.make_params(**{**case["findflanks_params"], **case["pairflanks_params"]})
    g, goff = pfh.genome_arrays(genome, packed.contig_names)
    return packed, params, g, goff, pfh.contig_ranks(packed.contig_names)


def _worker(rank, world, port, out_dir, cap, case_name, sharding):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    ora = Oracle()
    packed, params, g, goff, ranks = _inputs(case_name)
    h = mdist.halo(params.min_distance_to_mate, params.max_direct_repeat_length)
    plan = mdist.plan_shards(packed, world, sharding, h)[rank]
    rd = ora.reads(packed.take_ranges(plan.ranges))
    tab, seqs = mdist.findflanks_sharded(ora, rd, plan, params, device="cpu", cap=cap)
    pairs, strs = mdist.pairflanks_sharded(ora, rd, plan, params, tab, seqs, ranks, g, goff, device="cpu", cap=cap)
    with open(os.path.join(out_dir, "rank%d.pkl" % rank), "wb") as fh:
        pickle.dump((tab, seqs, pairs, strs, plan.n_reads()), fh)
    dist.destroy_process_group()


@pytest.mark.parametrize("case_name,sharding,cap", [("sim_b", "coordinate", mdist.GATHER_CAP),
                                                    ("sim_b", "coordinate", 64),   # 64 bytes: the tables do not fit, the second all-gather runs
                                                    ("multi", "coordinate", mdist.GATHER_CAP),
                                                    ("multi", "contig", mdist.GATHER_CAP)])
def test_sharded_stages_merge_to_the_unsharded_tables(tmp_path, oracle, case_name, sharding, cap):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), cap, case_name, sharding), nprocs=world, join=True)
    got = [pickle.load(open(os.path.join(tmp_path, "rank%d.pkl" % r), "rb")) for r in range(world)]
    packed, params, g, goff, ranks = _inputs(case_name)
    R = oracle.reads(packed)
    want = oracle.run_findflanks(R, params)
    wantp = oracle.run_pairflanks(R, params, want["flanks"], want["seqs"], ranks, g, goff)
    wp, ws = mdist._compact(wantp["pairs"], wantp["strs"], mdist._PAIR_STR)
    assert len(want["flanks"]) > 0 and len(wp) > 0
    for tab, seqs, pairs, strs, n_reads in got:                       # every rank holds both merged tables
        assert tab.tobytes() == want["flanks"].tobytes() and seqs == want["seqs"]
        assert pairs.tobytes() == wp.tobytes() and strs == ws
    assert all(0 < n < packed.n_reads for *_, n in got)              # and computed on its share of the reads only


def test_shard_helpers():
    assert mdist.shard_samples(5, 2, 0) == [0, 2, 4] and mdist.shard_samples(5, 2, 1) == [1, 3]
    bins = mdist.shard_contigs([100, 5, 50, 40, 7], 2)
    assert sorted(sum(bins, [])) == [0, 1, 2, 3, 4]
    assert mdist.halo(22, 20) == 69
    assert mdist.halo(5, 100) == 102


@pytest.mark.parametrize("sharding", ["coordinate", "contig"])
@pytest.mark.parametrize("world", [1, 2, 3, 5, 8, 16])
def test_every_position_has_one_owner(world, sharding):
    packed, _, _ = synth_packed(contig_lens=(60_000, 9_000, 2_000, 30_000), coverage=20, n_insertions=4, seed=2)
    plans = mdist.plan_shards(packed, world, sharding)
    assert len(plans) == world
    for c, clen in enumerate(packed.contig_len):
        pos = np.arange(-1, int(clen) + 1)
        owners = sum(p.owns(np.full(pos.shape, c), pos).astype(int) for p in plans)
        assert np.all(owners == 1), (c, world)
    # a plan's reads hold everything that can overlap what it owns (+ halo)
    h = mdist.halo()
    for p in plans:
        for c, a, b in p.pieces:
            r0, r1 = int(packed.contig_read_start[c]), int(packed.contig_read_start[c + 1])
            idx = np.arange(r0, r1)
            need = idx[(packed.ref_end[r0:r1] > a - h - 1) & (packed.ref_start[r0:r1] < b + h + 2)]
            have = np.zeros(packed.n_reads, bool)
            for lo, hi in p.ranges:
                have[lo:hi] = True
            assert have[need].all()


def test_take_ranges_equals_slices(oracle):
    """Reads of several ranges as one PackedReads: same arrays as the single slices, links re-based across ranges."""
    packed, _, _ = synth_packed(**MULTI, seed=3)
    n = packed.n_reads
    ranges = [(10, 5000), (5000 + 17, 90_000), (n - 4000, n)]
    sub = packed.take_ranges(ranges)
    idx = np.concatenate([np.arange(lo, hi) for lo, hi in ranges])
    for k in ("ref_start", "ref_end", "contig_id", "qlen", "first_op_len", "last_op_len", "mapq", "flags"):
        assert np.array_equal(sub.a[k], packed.a[k][idx]), k
    new_of_old = np.full(n, -1, np.int64)
    new_of_old[idx] = np.arange(idx.shape[0])
    link = packed.link[idx].astype(np.int64)
    assert np.array_equal(sub.link, np.where(link >= 0, new_of_old[np.maximum(link, 0)], -1))
    assert (sub.link >= 0).sum() > 0
    for c in range(len(packed.contig_names) + 1):
        assert sub.contig_read_start[c] == (packed.contig_id[idx] < c).sum()
    for new, old in list(enumerate(idx))[::997]:
        assert sub.pool_sequence(new) == packed.pool_sequence(int(old))
    # block table of complex reads
    k_new = np.flatnonzero(sub.flags & 16)
    assert np.array_equal(sub.cx_read, k_new)
    for j in range(0, len(k_new), 37):
        old = int(idx[k_new[j]])
        jo = int(np.searchsorted(packed.cx_read, old))
        assert np.array_equal(sub.blk_start[sub.cx_off[j]:sub.cx_off[j + 1]], packed.blk_start[packed.cx_off[jo]:packed.cx_off[jo + 1]])

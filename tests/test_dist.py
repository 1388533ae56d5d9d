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


def test_c_table_loops_equal_the_numpy_statement(monkeypatch):
    """mfk_flanks_owned / mfk_flank_runs_owned / mfk_pairs_owned / mfk_flanks_merge / mfk_pairs_merge (libmfk.so, host code, no
    GPU needed) against the numpy loops of mustache_amd/dist.py on random tables: several contigs, pieces that cut a contig,
    contigs a shard owns nothing of, None strings (negative lengths), empty tables, parts in and out of table order."""
    import numpy as np
    from mustache_amd import dist as md, lib
    assert md._tables() is not None, "libmfk.so must be built"
    rng = np.random.default_rng(8)

    def random_flanks(n, nc):
        t = np.zeros(n, dtype=lib.FLANK_DTYPE)
        t["contig_id"] = np.sort(rng.integers(0, nc, n))
        t["pos"] = rng.integers(-1, 5000, n)
        order = np.lexsort((t["pos"], t["contig_id"]))
        t = t[order]
        t["orient"] = rng.choice([5, 3], n)
        t["seq_len"] = rng.integers(0, 60, n)
        perm = rng.permutation(n)                                     # strings scattered in the blob
        off = np.zeros(n, dtype=np.int64)
        off[perm] = np.concatenate([[0], np.cumsum(t["seq_len"][perm])[:-1]])
        t["seq_off"] = off
        return t, bytes(rng.integers(65, 70, int(t["seq_len"].sum()) + 3, dtype=np.uint8))

    def random_pairs(n, nc, nsel):
        p = np.zeros(n, dtype=lib.PAIR_DTYPE)
        p["contig_id"] = np.sort(rng.integers(0, nc, n))
        p["pos_5p"] = rng.integers(0, 5000, n)
        p["pos_3p"] = p["pos_5p"] - rng.integers(1, 20, n)
        p["index_5p"], p["index_3p"] = rng.integers(0, nsel, n), rng.integers(0, nsel, n)
        o = 0
        for fo, fl in md._PAIR_STR:
            p[fl] = rng.integers(-1, 30, n)
        lens = np.stack([np.maximum(p[fl], 0) for _, fl in md._PAIR_STR], 1).reshape(-1)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).reshape(n, 4)
        for j, (fo, _) in enumerate(md._PAIR_STR):
            p[fo] = offs[:, j]
        return p, bytes(rng.integers(65, 70, int(lens.sum()) + 1, dtype=np.uint8))

    def both(fn):
        a = fn()
        monkeypatch.setenv("MFK_NUMPY_MERGE", "1")
        b = fn()
        monkeypatch.delenv("MFK_NUMPY_MERGE")
        return a, b

    for n, nc in [(0, 3), (1, 1), (300, 4), (2000, 7)]:
        plan = md.ShardPlan([(1, 700, 5001), (2, -1, 5001), (3, -1, 1200)] if nc > 3 else [(0, -1, 2500)], [(0, 1)])
        tab, seqs = random_flanks(n, nc)
        (r1, s1), (r2, s2) = both(lambda: md._own_flanks({"flanks": tab, "seqs": seqs}, plan))
        assert r1.tobytes() == r2.tobytes() and s1 == s2 and (n < 100 or 0 < len(r1) < n)
        for reach in (0, 21):
            a, b = both(lambda: md._flank_rows_for(plan, tab, reach))
            assert np.array_equal(a, b)
        sel = md._flank_rows_for(plan, tab, 21)
        if len(sel) == 0:
            continue
        pr, strs = random_pairs(max(n // 2, 1), nc, len(sel))

        class FakeBinding:
            def run_pairflanks(self, *a, **k):
                return {"pairs": pr.copy(), "strs": strs, "stats": np.zeros(16, np.int64)}
        params = lib.make_params()
        (p1, t1), (p2, t2) = both(lambda: md.shard_pairflanks(FakeBinding(), None, plan, params, tab, seqs, np.arange(nc), None, None))
        assert p1.tobytes() == p2.tobytes() and t1 == t2
        # merges: three parts in table order, then the same parts in another order (whole-contig shards)
        cuts = [0, len(r1) // 3, 2 * len(r1) // 3, len(r1)]
        parts = [md._compact(r1[cuts[i]:cuts[i + 1]], s1, md._FLANK_STR) for i in range(3)]
        for ps in (parts, parts[::-1]):
            (m1, b1), (m2, b2) = both(lambda: md.merge_flanks(ps, compacted=True))
            assert m1.tobytes() == m2.tobytes() and b1 == b2
        ranks = rng.permutation(nc).astype(np.int32)
        order = np.lexsort((p1["pos_3p"], p1["pos_5p"], ranks[p1["contig_id"]]))
        sp, st = md._compact(md._fast_take(p1, order), t1, md._PAIR_STR)
        cuts = [0, len(sp) // 2, len(sp)]
        pparts = [md._compact(sp[cuts[i]:cuts[i + 1]], st, md._PAIR_STR) for i in range(2)]
        for ps in (pparts, pparts[::-1]):
            (m1, b1), (m2, b2) = both(lambda: md.merge_pairs(ps, ranks, compacted=True))
            assert m1.tobytes() == m2.tobytes() and b1 == b2

"""Multi-process layer on CPU: gloo, world_size 2.  Sharding must not change results: coordinate shards with the
3 x mindist halo, run separately (CPU oracle) and merged with the all-gather, equal the unsharded table."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mustache_amd import dist as mdist
from mustache_amd import lib
from mustache_amd.pack import pack_reads
from tests import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, cap):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    ora = Oracle()
    case = util.load_json("sim_b")
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    params = lib.make_params(**case["findflanks_params"])
    h = mdist.halo(params.min_distance_to_mate, params.max_direct_repeat_length)
    own = mdist.coordinate_shards(packed, 0, world)[rank]
    sub = mdist.slice_for_range(packed, 0, own, h)
    res = ora.run_findflanks(ora.reads(sub), params)
    mine = mdist.owned(res["flanks"], own, contig=0)
    # re-base the sequence bytes of the owned rows
    seqs, off = [], 0
    mine = mine.copy()
    for row in mine:
        s = res["seqs"][int(row["seq_off"]):int(row["seq_off"]) + int(row["seq_len"])]
        row["seq_off"] = off
        off += len(s)
        seqs.append(s)
    rows, blob = mdist.allgather_flanks(mine, b"".join(seqs), device="cpu", cap=cap)
    if rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), rows)
        with open(os.path.join(out_dir, "seqs.bin"), "wb") as fh:
            fh.write(blob)
    dist.destroy_process_group()


@pytest.mark.parametrize("cap", [mdist.GATHER_CAP, 64])   # 64 bytes: the table does not fit, the second all-gather runs
def test_coordinate_shards_merge_to_the_unsharded_table(tmp_path, oracle, cap):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), cap), nprocs=world, join=True)
    rows = np.load(os.path.join(tmp_path, "rows.npy"))
    blob = open(os.path.join(tmp_path, "seqs.bin"), "rb").read()
    case = util.load_json("sim_b")
    rs = util.case_readset(case)
    packed = pack_reads(rs.contigs, rs, pool="all")
    want = oracle.run_findflanks(oracle.reads(packed), lib.make_params(**case["findflanks_params"]))
    assert len(rows) == len(want["flanks"]) > 0
    for name in lib.FLANK_DTYPE.names:
        if name != "seq_off":
            assert np.array_equal(rows[name], want["flanks"][name]), name
    got_seqs = [blob[int(r["seq_off"]):int(r["seq_off"]) + int(r["seq_len"])] for r in rows]
    want_seqs = [want["seqs"][int(r["seq_off"]):int(r["seq_off"]) + int(r["seq_len"])] for r in want["flanks"]]
    assert got_seqs == want_seqs


def test_shard_helpers():
    assert mdist.shard_samples(5, 2, 0) == [0, 2, 4] and mdist.shard_samples(5, 2, 1) == [1, 3]
    bins = mdist.shard_contigs([100, 5, 50, 40, 7], 2)
    assert sorted(sum(bins, [])) == [0, 1, 2, 3, 4]
    assert mdist.halo(22, 20) == 69
